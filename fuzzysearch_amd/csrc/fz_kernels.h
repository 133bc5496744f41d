// fz_kernels.h — hand-written HIP kernels for gfx950 (MI355X, CDNA4).  Included by fzhip.hip.
//
//   fz_scan_kernel    ONE streaming pass over the resident sequence:
//     K1 filter   every byte offset is tested against all G n-gram blocks at once (replaces G x
//                 search_exact_byteslike passes, _common.c:75-102 / memmem.c:92-160): a one-op
//                 hash of the first min(L,8) window bytes (v_alignbyte / v_mad_u32_u24) selects one
//                 of 32 slots of a table in LDS that holds the blocks' hashes (ds_read_b32: one slot
//                 per bank, so the lookup is bank-conflict free by construction), one v_xor
//                 compares, v_min3 accumulates, one ballot per 4 offsets: the cost does not depend
//                 on G; survivors go through a per-wave LDS queue to an exact re-check;
//     K2 verify   confirmed hits wait in a per-wave LDS staging area and are verified 64 at a
//                 time, one lane per hit, inside the same kernel: the <= m+2k window bytes are
//                 fetched once into LDS, then the bounded edit-distance expansion right and left
//                 (c_expand_*, _levenshtein_ngrams.pyx:9-154) or the Hamming count
//                 (_substitutions_only_ngrams_template.h:103-121) runs out of LDS / registers.
//                 Only match records leave the chip.
//                 (Measured and rejected in round 2, DESIGN.md §4: a persistent grid whose waves draw
//                 4 KiB chunks from ticket counters, and queueing only a fired group's position with
//                 the offset/block resolved at the flush — equal without candidates, 0.04 ms slower on
//                 DNA: every wave then flushes at the same time, at the end.)
//   fz_verify_kernel  the lane-per-candidate verification over a hit list in HBM, for parameter
//                 ranges that do not fit beside the filter (large m + 2k, budgets above 31).
//   fz_verify_wf_kernel  budgets 5..31: lane-per-DP-cell.  GW = 16 / 32 / 64 lanes own the 2k+1 band
//                 cells of one candidate's DP row; the left-neighbour recurrence is a prefix-min
//                 over the lanes (DPP row shifts), the upper neighbour one DPP shift.
//   fz_lp_kernel      one wave per work item: the reference's greedy candidate-set automata
//                 (generic search per n-gram hit; generic / Levenshtein linear-programming fallbacks
//                 tiled over the whole sequence), candidate lists in LDS, order preserved.
//   fz_hamming_kernel every window's Hamming distance (substitutions-only LP fallback).
//
// HBM-bound integer/byte work: no MFMA.  What matters (MI355X guide): 16-byte coalesced loads,
// >= 2048 workgroups' worth of loads in flight, no per-byte branching, n-gram constants in a
// 128-byte LDS table, wave-uniform rare paths, ONE global atomic per bulk append (a single counter
// word sustains only ~90 atomics/us chip-wide), no agent-scope fences.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <type_traits>
#include "fz_device.h"

#define FZ_FILTER_THREADS 256
#define FZ_WAVES_PER_BLOCK (FZ_FILTER_THREADS / 64)
#define FZ_FILTER_ROWS 4                                   // 16-byte rows per thread per tile
#define FZ_ROW_BYTES (FZ_FILTER_THREADS * 16)              // 4 KiB
#define FZ_TILE_BYTES (FZ_ROW_BYTES * FZ_FILTER_ROWS)      // 16 KiB (4 rows)
#define FZ_TILE_BITS 14                                    // log2(FZ_TILE_BYTES)
#define FZ_TITER_MAX ((1u << (32 - FZ_TILE_BITS - FZ_BLK_BITS)) - 1) // tile iterations a queue code can carry
#define FZ_PAD_FRONT 256                                   // zero bytes before buf[0]
#define FZ_PAD_BACK 256                                    // zero bytes the kernels may over-read (halo loads, whole 16-byte window pieces)
#ifndef FZ_GROUP
#define FZ_GROUP 0                                         // lab knob: force 4 or 8 byte offsets per wave-uniform branch (0: by n-gram length)
#endif
#define FZ_WF_COMPACT_MIN 8192ull                          // fz_verify_wf_kernel: record slots above which records are appended, not slotted
#define FZ_QCAP 256                                        // fast-hit queue entries per wave
#ifndef FZ_LUT_BITS
#define FZ_LUT_BITS 5                                      // 32 slots: one per LDS bank (6 = the round-1 table, 2-way conflicts)
#endif
#define FZ_LUT_SLOTS (1u << FZ_LUT_BITS)                   // slots of the block-hash table
#define FZ_LUT_BYTES (FZ_LUT_SLOTS * 4u)
#define FZ_TABLE_BYTES (2u * FZ_LUT_BYTES + 32u)           // the hash table + one dword per slot: the block that lives there,
                                                           // four dwords of the pooled flush, the workgroup's tile walk (below)
#define FZ_WALK_LDS (2u * FZ_LUT_BYTES + 16u)              // LDS address of {first tile lo, hi, tile stride}: the scan kernel's
                                                           // dynamic LDS starts at address 0
                                                           // + the four waves' final queue fills (pooled last flush)
                                                           // (same byte offset as the hash: no address arithmetic in the rare path)
#define FZ_FLAG_DUP_HASHES 1u                              // FzScanArgs.flags: two blocks of the launch have the same hash
#define FZ_FLAG_FOLD 4u                                    // per-hit automaton: fold the matches of a hit into (hull, best match) pairs on
                                                           // the device (consolidate_overlapping_matches, common.py:145-189, first stage)
#define FZ_FLAG_ANY 2u                                     // has_near_match_*: the caller only asks WHETHER a record exists — work that starts
                                                           // after the first record has been counted is skipped
#if FZ_LUT_BITS == 5
#define FZ_LUT_ADDR_MASK_STR "0x7c"                        // (FZ_LUT_SLOTS - 1) * 4: byte address of a slot
#else
#define FZ_LUT_ADDR_MASK_STR "0xfc"
#endif

// 32-bit little-endian window starting `b` bytes into the 64-bit value hi:lo (v_alignbyte_b32).
__device__ __forceinline__ uint32_t fz_win(uint32_t lo, uint32_t hi, int b) {
    return b == 0 ? lo : __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)b);
}
#define FZ_WIN(w, o) fz_win((w)[(o) >> 2], (w)[((o) >> 2) + 1], (o) & 3)

typedef __attribute__((address_space(3))) const uint32_t FzLdsU32;   // LDS seen through a raw 32-bit LDS address
typedef __attribute__((address_space(3))) uint8_t FzLdsU8;

__device__ __forceinline__ uint32_t fz_lane() { return threadIdx.x & 63u; }

// The pattern into LDS: from the kernel-argument block, or (patterns longer than FZ_MAX_M, fz_verify_big_kernel's
// callers) from HBM.  Two loops under one uniform branch, NOT a select per byte between the two sources: a select
// between an argument-block address and a global one makes hipcc copy the whole 1.5 KB argument struct to scratch
// memory (1472 bytes per lane in every kernel that did it; the exact-search scan ran 3.4x slower).
__device__ __forceinline__ void fz_copy_pattern(uint8_t *dst, const FzScanArgs &a, uint32_t tid, uint32_t nthreads) {
    if (a.pat_g) {
        const uint8_t *pg = reinterpret_cast<const uint8_t *>(a.pat_g);
        for (uint32_t i = tid; i < a.m; i += nthreads) dst[i] = pg[i];
    } else {
        for (uint32_t i = tid; i < a.m; i += nthreads) dst[i] = a.pat[i];
    }
}

__device__ __forceinline__ uint32_t fz_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Make this wave's earlier LDS writes visible to its own later LDS reads (wave-synchronous code:
// no other wave shares these LDS regions, so no s_barrier is needed).
__device__ __forceinline__ void fz_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// 32-bit window at an arbitrary (unaligned) local byte position: two aligned dword loads.
__device__ __forceinline__ uint32_t fz_load_win(const uint8_t *__restrict__ buf, int64_t local) {
    const int64_t base = local & ~(int64_t)3;
    const uint32_t lo = *reinterpret_cast<const uint32_t *>(buf + base);
    const uint32_t hi = *reinterpret_cast<const uint32_t *>(buf + base + 4);
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(local & 3));
}

// (The lab instrumentation of rounds 2 - 5 — device time stamps per workgroup phase / per n-gram hit, kernels with a part of
// the candidate handling left out, the miscompiled loop shape of fz_levlp_step_slots — is not in this file: benchmarks/
// lab_patches/lab_instrumentation.patch re-inserts it, benchmarks/lab_build.sh applies it to a scratch copy.)

// Per-wave LDS areas, carved from dynamic LDS by fz_wave_lds() / fz_wave_lds_pref().
struct FzWaveLds {
    uint32_t *queue;      // [qcap]  fast hits: tile-local offset | block << 14 | tile iteration << (14 + FZ_BLK_BITS)
    uint32_t *win;        // staged layout:     [win_dwords * vlanes]  dword d of the hit in slot l at d*vlanes+l
                          // prefetched layout: [win_pieces][qcap][16 bytes]  piece c of queue entry e at (c*qcap+e)*16
    uint16_t *scores;     // [band_w * vlanes]  ring of DP score slots (slot s of lane l at s*vlanes+l); none when prefetched
    uint32_t win_lds;     // LDS byte address of `win` (the LDS-DMA destination base; wave-uniform)
};

__host__ __device__ inline uint32_t fz_wave_lds_bytes(uint32_t win_dwords, uint32_t band_w, uint32_t vlanes,
                                                      bool with_queue) {
    uint32_t b = 0;
    if (with_queue) b += FZ_QCAP * 4;
    b += win_dwords * vlanes * 4;
    b += ((band_w * vlanes * 2) + 15u) & ~15u;
    return b;
}

// Fused lane-per-cell verification: 64 / gw contiguous byte windows per wave (window + 16 bytes of slack each), in dwords.
__host__ __device__ inline uint32_t fz_wf_fused_dwords(uint32_t win_dwords, uint32_t gw) { return (64u / gw) * (win_dwords + 4u); }

__device__ __forceinline__ FzWaveLds fz_wave_lds(uint8_t *base, uint32_t wave, uint32_t win_dwords, uint32_t band_w,
                                                 uint32_t vlanes, bool with_queue) {
    uint8_t *p = base + (size_t)wave * fz_wave_lds_bytes(win_dwords, band_w, vlanes, with_queue);
    FzWaveLds w;
    w.queue = nullptr;
    if (with_queue) { w.queue = reinterpret_cast<uint32_t *>(p); p += FZ_QCAP * 4; }
    w.win = reinterpret_cast<uint32_t *>(p); p += win_dwords * vlanes * 4;
    w.scores = reinterpret_cast<uint16_t *>(p);
    w.win_lds = 0;
    return w;
}

// Fused in-memory scan: every queue entry owns win_pieces 16-byte pieces of its sequence window, filled by
// LDS-DMA while the scan goes on (fz_prefetch_windows).
__host__ __device__ inline uint32_t fz_wave_lds_pref_bytes(uint32_t qcap, uint32_t win_pieces) {
    return qcap * 4u + win_pieces * qcap * 16u;
}

// `base_lds` = LDS byte address of `base` (the scan kernel's dynamic LDS starts at address 0).
__device__ __forceinline__ FzWaveLds fz_wave_lds_pref(uint8_t *base, uint32_t base_lds, uint32_t wave, uint32_t qcap,
                                                      uint32_t win_pieces) {
    const uint32_t off = wave * fz_wave_lds_pref_bytes(qcap, win_pieces);
    FzWaveLds w;
    w.queue = reinterpret_cast<uint32_t *>(base + off);
    w.win = reinterpret_cast<uint32_t *>(base + off + qcap * 4u);
    w.scores = nullptr;
    w.win_lds = base_lds + off + qcap * 4u;
    return w;
}

// LDS-resident accessors used by fz_verify_* on the GPU.
struct FzLdsScores {
    uint16_t *base;                                        // already offset by the lane
    uint32_t stride;                                       // = vlanes
    __device__ __forceinline__ uint32_t get(uint32_t slot) const { return base[slot * stride]; }
    __device__ __forceinline__ void set(uint32_t slot, uint32_t v) { base[slot * stride] = (uint16_t)v; }
};
struct FzLdsWindow {
    const uint8_t *base;                                   // lane's dword 0, as bytes
    uint64_t wbase;                                        // global index of byte 0 of the window
    uint32_t stride4;                                      // = vlanes * 4: bytes between a lane's consecutive dwords
    __device__ __forceinline__ uint8_t at(uint64_t gidx) const {
        const uint32_t off = (uint32_t)(gidx - wbase);
        return base[(off >> 2) * stride4 + (off & 3u)];
    }
};

// A window that LDS-DMA laid down: 16-byte piece c of the entry at base + c * rstride.
struct FzDmaWindow {
    const uint8_t *base;                                   // piece 0 of this entry
    uint64_t wbase;                                        // global index of byte 0 of the window
    uint32_t rstride;                                      // = qcap * 16
    __device__ __forceinline__ uint8_t at(uint64_t gidx) const {
        const uint32_t off = (uint32_t)(gidx - wbase);
        return base[(off >> 4) * rstride + (off & 15u)];
    }
    // aligned dword at window byte offset o (o % 4 == 0); offsets past the staged pieces read other LDS
    // bytes (or 0 beyond the allocation): callers only use the bytes they know to be staged
    __device__ __forceinline__ uint32_t dword(uint32_t o) const {
        return *reinterpret_cast<const uint32_t *>(base + (o >> 4) * rstride + (o & 15u));
    }
    // byte at window offset o
    __device__ __forceinline__ uint32_t byte(uint32_t o) const { return base[(o >> 4) * rstride + (o & 15u)]; }
};

// The two whole-pattern Peq tables of the bit-vector verification in LDS (fz_device.h: fz_verify_lev_bits): [side][256]
// words of NW x 8 bytes, forward table first.
#define FZ_PEQ_BYTES(NW) (2u * 256u * (FZ_BITS_WIDTH(NW) / 8u))
template <int NW>
struct FzPeqLds {
    typedef typename FzBitsWord<NW>::T T;
    typedef __attribute__((address_space(3))) const T LdsT;
    uint32_t tabs;                                         // LDS byte address of the forward table
    // a table's handle = its LDS address: a word's address is one v_lshl_add_u32
    __device__ __forceinline__ uint32_t table(uint32_t side) const { return tabs + side * (256u * (FZ_BITS_WIDTH(NW) / 8u)); }
    __device__ __forceinline__ T at(uint32_t tab, uint32_t ch) const {
        return *(LdsT *)(uintptr_t)(tab + ch * (FZ_BITS_WIDTH(NW) / 8u));
    }
};

// N dwords of the byte string that starts at byte offset `off` of a dword-readable source: rd(o) = the aligned
// dword at byte offset o.  One v_alignbyte per dword brings the string to byte 0 of the result.
template <int N, class Rd>
__device__ __forceinline__ FzBytes<N> fz_load_bytes(Rd rd, uint32_t off) {
    const uint32_t a0 = off & ~3u, sh = off & 3u;
    uint32_t raw[N + 1];
#pragma unroll
    for (int q = 0; q <= N; ++q) raw[q] = rd(a0 + 4u * (uint32_t)q);
    FzBytes<N> b;
#pragma unroll
    for (int q = 0; q < N; ++q) b.r[q] = __builtin_amdgcn_alignbyte(raw[q + 1], raw[q], sh);
    return b;
}

// Start of the staged window of the hit (block starting at s, index idx) inside the segment that starts at sa:
// the dword-aligned buffer position at or below max(sa, idx - s - k).
__device__ __forceinline__ uint64_t fz_window_lo(const FzScanArgs &a, uint64_t idx, uint32_t s, uint64_t sa) {
    const uint64_t reach = (uint64_t)s + a.k;
    uint64_t wlo = idx - sa > reach ? idx - reach : sa;
    if (wlo < a.geom.buf_off) wlo = a.geom.buf_off;
    return wlo;
}
__device__ __forceinline__ uint64_t fz_window_base(const FzScanArgs &a, uint64_t wlo) {
    return a.geom.buf_off + ((wlo - a.geom.buf_off) & ~(uint64_t)3);
}

// Asynchronous 16-byte copy global -> LDS (global_load_lds_dwordx4): every active lane l copies 16 bytes from
// ITS address gsrc to LDS byte address lds_dst + 16 * l (lds_dst wave-uniform, goes through M0).  No VGPR is
// written; completion is tracked by vmcnt like any load.  hipcc does not count this statement in its own
// s_waitcnt bookkeeping: its vmcnt(N) waits then cover MORE than it intended (vmcnt(N) = "at most N still
// in flight", and loads retire in order), never less; the consumer of the LDS bytes waits vmcnt(0) itself.
__device__ __forceinline__ void fz_glds16(const uint8_t *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ unsigned long long fz_bcast64(unsigned long long v) {
    // (the builtin returns int: widen through uint32_t, or a low word with bit 31 set sign-extends into the high word)
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
           (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// _substitutions_only_ngrams_template.h:103-121 for the hits of a wave, four characters per step: the window's Hamming
// distance to the WHOLE pattern (the n-gram's own positions agree — the caller has confirmed the n-gram — and add nothing),
// text dwords brought to the pattern's alignment by v_alignbyte, nonzero bytes of the xor counted by one v_bcnt.  The loop is
// wave-uniform (m / 4 steps, left when no lane is within the budget any more): fz_verify_subs's per-lane skip over the
// n-gram and per-lane early return compile to ~90 scalar instructions per character and lane group on dense candidates
// (1 GiB DNA, m = 20, 4 substitutions: 1.9e9 SALU instructions).  -> dist = the Hamming distance where it is <= k.
__device__ __forceinline__ bool fz_verify_subs_wave(const FzDmaWindow &t, const uint8_t *pat_lds, uint32_t m, uint32_t k, uint32_t L,
                                                    uint32_t s, uint64_t idx, bool valid, FzRec &rec) {
    const uint32_t off = valid ? (uint32_t)(idx - s - t.wbase) : 0u;
    const uint32_t a0 = off & ~3u, sh = off & 3u;
    uint32_t prev = t.dword(a0), nd = 0;
    for (uint32_t q = 0; q < m; q += 4u) {
        const uint32_t next = t.dword(a0 + q + 4u);
        uint32_t d = __builtin_amdgcn_alignbyte(next, prev, sh) ^ *reinterpret_cast<const uint32_t *>(pat_lds + q);
        if (q + 4u > m) d &= (1u << (8u * (m - q))) - 1u;              // (uniform: the pattern's last, partial dword)
        const uint32_t nz = (((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) & 0x80808080u;   // bit 7 of every nonzero byte
        nd += (uint32_t)__popc(nz);
        prev = next;
        if ((q & 4u) && !__ballot(valid && nd <= k)) break;
    }
    rec.l = s; rec.r = m - s - L; rec.dist = nd; rec.aux = 0;
    return valid && nd <= k;
}

// ---------------------------------------------------------------------------------------------
// Wave-level verification of up to 64 candidates: each valid lane owns one (hit = block | idx), the segment it
// is verified in and a slot vl of the window area.
//  1. staged form (PREF = false): every lane fetches the <= m + 2k window bytes around its candidate into LDS
//     with independent aligned dword loads (slot vl < a.vlanes);
//     prefetched form (PREF = true, the fused in-memory scan): the window of queue entry vl is already in LDS —
//     fz_prefetch_windows requested it by LDS-DMA right after the tile that produced the hit, while its cache
//     lines were still in L2, and the scan went on meanwhile.  (Measured, round 2: the fetch at flush time, one
//     exposed HBM round trip per wave at the end of its life, cost 0.019 of the 0.226 ms of the headline scan;
//     the DP itself cost nothing measurable.)
//  2. the n-gram is confirmed exactly (the filter only compared a hash of its first min(L, 8) bytes), then the
//     reference's per-hit logic (fz_verify_lev / fz_verify_subs) runs out of LDS,
//  3. the wave appends its records with ONE global atomic.
// Returns the number of exactly-confirmed n-gram hits (wave-uniform, statistics).
//  BITS = 1 / 2 (prefetched form, Levenshtein): the two expansions run as bit-vector columns on 64 / 128-bit words
//  (fz_verify_lev_bits) with the Peq tables at `peq_tabs` in LDS, instead of the register band.
template <int MAXK, bool PREF, int BITS = 0>
__device__ __forceinline__ uint32_t fz_wave_verify(const uint8_t *__restrict__ buf, const FzScanArgs &a,
                                                   const uint8_t *pat_lds, const FzWaveLds &w, uint32_t vl,
                                                   uint64_t hit, const FzSeg &sg, bool valid,
                                                   FzRec *__restrict__ recs, unsigned long long *__restrict__ counters,
                                                   const uint8_t *pref_win = nullptr, const uint8_t *peq_tabs = nullptr) {
    static_assert(BITS == 0 || PREF, "the bit-vector form verifies prefetched windows");     // (BITS = -1: Hamming count only)
    const uint32_t lane = fz_lane();
    const uint32_t g = fz_hit_block(hit);
    const uint64_t idx = fz_hit_index(hit);
    const uint32_t s = g * a.L;
    // window [wlo, whi) in global coordinates; every byte the verification touches lies inside it
    const uint64_t wlo = fz_window_lo(a, idx, s, sg.sa);
    const uint64_t wbase = fz_window_base(a, wlo);
    if constexpr (!PREF) {
        uint64_t whi = idx - s + a.m + a.k;
        const uint64_t lim = a.geom.buf_off + a.geom.buf_len;
        if (whi > lim) whi = lim;
        if (whi > sg.se) whi = sg.se;
        const uint32_t nd = valid ? (uint32_t)((whi - wbase + 3) >> 2) : 0u;
        const int64_t lbase = (int64_t)(wbase - a.geom.buf_off);
        // eight loads in flight per lane, then eight LDS stores (a load-store loop would pay the HBM/L2
        // round trip once per dword: 8 - 21 serial round trips per flush)
        for (uint32_t d0 = 0; d0 < a.win_dwords; d0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j)
                v[j] = (d0 + j < nd) ? *reinterpret_cast<const uint32_t *>(buf + lbase + (int64_t)(d0 + j) * 4) : 0u;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j)
                if (d0 + j < nd) w.win[(d0 + j) * a.vlanes + vl] = v[j];
        }
        fz_wave_lds_sync();
    }
    FzRec rec;
    bool ok = false;
    auto run = [&](const auto &t) {
        auto prd = [&](uint32_t o) -> uint32_t { return *reinterpret_cast<const uint32_t *>(pat_lds + o); };
        if constexpr (PREF) {
            // exact n-gram test on registers: 8 bytes of text and pattern, masked to min(L, 8) (uniform), the
            // rest (L > 8) byte by byte
            const uint32_t off = (uint32_t)(idx - wbase);
            const FzBytes<2> tn = fz_load_bytes<2>([&](uint32_t o) { return t.dword(o); }, off);
            const FzBytes<2> pn = fz_load_bytes<2>(prd, s);
            const uint32_t m0 = a.L >= 4 ? 0xffffffffu : (1u << (8u * a.L)) - 1u;
            const uint32_t m1 = a.L >= 8 ? 0xffffffffu : a.L > 4 ? (1u << (8u * (a.L - 4u))) - 1u : 0u;
            if ((((tn.r[0] ^ pn.r[0]) & m0) | ((tn.r[1] ^ pn.r[1]) & m1)) != 0) valid = false;
            if (a.L > 8 && valid) {
                const uint8_t *ng = pat_lds + s;
                for (uint32_t b = 8; b < a.L; ++b)
                    if (ng[b] != t.at(idx + b)) { valid = false; break; }
            }
        } else {
            if (valid) {
                const uint8_t *ng = pat_lds + s;
                for (uint32_t b = 0; b < a.L; ++b)
                    if (ng[b] != t.at(idx + b)) { valid = false; break; }
            }
        }
        const uint32_t confirmed = (uint32_t)__popcll(__ballot(valid));
        if constexpr (BITS < 0) {                            // the substitutions-only form alone (fz_scan_kernel<..., 3>)
            ok = fz_verify_subs_wave(t, pat_lds, a.m, a.k, a.L, s, idx, valid, rec);
        } else if constexpr (BITS != 0) {
            // every lane takes part: the loop's control is wave-uniform (lanes without a candidate idle in it)
            const FzPeqLds<BITS> peq{(uint32_t)(uintptr_t)(FzLdsU8 *)peq_tabs};
            ok = fz_verify_lev_bits<BITS>(peq, [&](uint32_t o) -> uint32_t { return t.byte(o); }, wbase, sg.sa, sg.se, a.m, a.k,
                                          a.L, s, idx, valid, rec);
        } else if (a.mode == FZ_MODE_LEV) {
            FzLdsScores sc{w.scores + (PREF ? 0u : vl), a.vlanes};
            if (valid) ok = fz_verify_lev<MAXK>(sc, t, sg.sa, sg.se, pat_lds, a.m, a.k, a.L, s, idx, rec);
        } else if (valid) {
            ok = fz_verify_subs(t, pat_lds, a.m, a.k, a.L, s, idx, rec);
        }
        return confirmed;
    };
    uint32_t confirmed;
    if constexpr (PREF) {
        confirmed = run(FzDmaWindow{pref_win, wbase, a.qcap * 16u});      // piece 0 of the lane's queue entry
    } else {
        confirmed = run(FzLdsWindow{reinterpret_cast<const uint8_t *>(w.win + vl), wbase, a.vlanes * 4u});
    }
    const unsigned long long mask = __ballot(ok);
    if (mask) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&counters[1], (unsigned long long)__popcll(mask));
        base = fz_bcast64(base);
        if (ok) {
            rec.key = hit;
            rec.aux = sg.j;
            const unsigned long long slot = base + fz_rank(mask);
            if (slot < a.rec_cap) recs[slot] = rec;
        }
    }
    fz_wave_lds_sync();
    return confirmed;
}

// Exact test of one (local position, block) candidate against the buffer in HBM (emit mode).
__device__ __forceinline__ bool fz_confirm(const uint8_t *__restrict__ buf, const FzScanArgs &a, uint32_t blk,
                                           uint64_t local) {
    if ((fz_load_win(buf, (int64_t)local) & a.mask1) != a.A[blk]) return false;
    if (a.L > 4 && (fz_load_win(buf, (int64_t)local + a.d2) & a.mask2) != a.B[blk]) return false;
    if (a.L > 8) {                                         // (two loops under a uniform branch: see fz_copy_pattern)
        if (a.pat_g) {
            const uint8_t *pg = reinterpret_cast<const uint8_t *>(a.pat_g) + a.s[blk];
            for (uint32_t b = 8; b < a.L; ++b)
                if (buf[local + b] != pg[b]) return false;
        } else {
            for (uint32_t b = 8; b < a.L; ++b)
                if (buf[local + b] != a.pat[a.s[blk] + b]) return false;
        }
    }
    return true;
}

#define FZ_HDR_WORDS 128                                   // 64-bit counters in the result header
#define FZ_HDR_TICKET 3                                    // counters[3]: ticket shards that are complete (final launch)
#define FZ_HDR_SHARD0 96                                   // counters[96 .. 111]: workgroups that finished, by blockIdx % 16
#define FZ_TICKET_SHARDS 16u

// End of the final kernel of a search: the LAST workgroup to get here copies the counters into
// host-visible memory (a.host_hdr), so the host needs no D2H copy command after the kernel (the
// records themselves are then written straight to pinned host memory as well).  Every thread of
// the workgroup must call it.
// No agent-scope fence on purpose: on this multi-XCD part __threadfence() writes back the XCD's L2,
// and one per workgroup made the scan 1.7x slower.  It is not needed either: everything the last
// workgroup reads was produced by agent-scope atomics (performed memory-side), __syncthreads() makes
// each wave wait for its own outstanding atomics (s_waitcnt vmcnt(0)) before the ticket is taken,
// and the counters are read back with agent-scope atomic loads.
// `flag` is one LDS dword the workgroup no longer needs (no static __shared__ here: it would move the
// dynamic LDS base off 0 and cost the scan an address add per table lookup).
__device__ __forceinline__ void fz_finish_launch(const FzScanArgs &a, unsigned long long *__restrict__ counters,
                                                 volatile uint32_t *flag, uint32_t participants = 0) {
    if (!a.host_hdr) return;
    if (participants == 0) participants = gridDim.x;       // workgroups that take a ticket (all, unless the caller says fewer)
    __syncthreads();                                       // all waves' counter atomics are complete, LDS is free
    if (threadIdx.x == 0) {
        // Two-level ticket: 16 shard words, then one word for the shards' last arrivals.  One word only sustains ~90
        // atomics per microsecond chip-wide, and the workgroups of a short launch all finish together (1536 of them on a
        // 64 MiB scan: 10 us of its 42 us were this queue).
        const uint32_t nshard = participants < FZ_TICKET_SHARDS ? participants : FZ_TICKET_SHARDS;
        const uint32_t r = blockIdx.x % nshard;
        const unsigned long long members = (participants - r + nshard - 1u) / nshard;   // workgroups b < participants with b % nshard == r
        bool last = atomicAdd(&counters[FZ_HDR_SHARD0 + r], 1ull) == members - 1ull;
        if (last) last = atomicAdd(&counters[FZ_HDR_TICKET], 1ull) == (unsigned long long)nshard - 1ull;
        *flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (*flag) {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.host_hdr);
        for (uint32_t i = threadIdx.x; i < FZ_HDR_WORDS; i += blockDim.x) {
            dst[i] = __hip_atomic_load(&counters[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ... and leaves the counters zeroed for the next search (no memset command on the stream)
            __hip_atomic_store(&counters[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// A wave-uniform value the compiler does not recognise as such (it then keeps it in a VGPR and updates it with
// VALU ops): reading it through v_readfirstlane pins it to an SGPR; on values it already knows to be uniform the
// call folds away.
__device__ __forceinline__ uint32_t fz_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Candidate code of the queue: tile-local byte offset (14 bits) | block (FZ_BLK_BITS bits) | tile iteration.
__device__ __forceinline__ uint32_t fz_code(uint32_t off, uint32_t blk, uint32_t titer) {
    return off | (blk << FZ_TILE_BITS) | (titer << (FZ_TILE_BITS + FZ_BLK_BITS));
}

// Queue entry -> (block of this launch, local byte position in the buffer).
__device__ __forceinline__ uint64_t fz_code_local(uint32_t code, uint32_t &blk) {
    blk = (code >> FZ_TILE_BITS) & (FZ_MAX_BLOCKS_PER_LAUNCH - 1u);
    // the workgroup's tile walk (first tile, stride) as the scan kernel left it in LDS
    FzLdsU32 *walk = (FzLdsU32 *)(uintptr_t)FZ_WALK_LDS;
    const uint64_t first = ((uint64_t)walk[1] << 32) | walk[0];
    const uint64_t tile = first + (uint64_t)(code >> (FZ_TILE_BITS + FZ_BLK_BITS)) * walk[2];
    return tile * (uint64_t)FZ_TILE_BYTES + (code & (FZ_TILE_BYTES - 1u));
}

// Fused in-memory scan: request the sequence windows of the queue entries [qf, qn) — a.win_pieces LDS-DMA
// copies of 16 bytes per entry, lane = entry, destination = the entry's slot of the window area.  Called
// after every tile that queued something, i.e. while the lines are still in L2 (the re-fetch at flush time
// used to be 10 % extra HBM traffic, profiles/r01_pmc_summary.json), and nothing waits for the data before
// the flush.  Entries that the range check will drop are fetched as well (their address is inside the
// buffer; cheaper than testing here).
__device__ __forceinline__ void fz_prefetch_windows(const uint8_t *__restrict__ buf, const FzScanArgs &a,
                                                    const FzWaveLds &w, uint32_t qf, uint32_t qn) {
    const uint32_t lane = fz_lane();
    // The queue entries were stored by this wave: a wave's LDS operations are performed in issue order, so the
    // reads below see them; only the compiler must not reorder.  (fz_wave_lds_sync() would be wrong here: its
    // workgroup-scope release is an s_waitcnt vmcnt(0), i.e. a wait for the rows just requested for the next tile.)
    asm volatile("" ::: "memory");
    for (uint32_t e0 = qf; e0 < qn; e0 += 64u) {
        const uint32_t e = e0 + lane;
        if (e < qn) {
            uint32_t blk;
            const uint64_t idx = a.geom.buf_off + fz_code_local(w.queue[e], blk);
            const uint64_t wbase = fz_window_base(a, fz_window_lo(a, idx, (a.g0 + blk) * a.L, 0));
            const uint8_t *src = buf + (wbase - a.geom.buf_off);
            for (uint32_t c = 0; c < a.win_pieces; ++c)
                fz_glds16(src + 16u * c, (uint32_t)__builtin_amdgcn_readfirstlane((int)(w.win_lds + (c * a.qcap + e0) * 16u)));
        }
    }
}

// The same for the entries a tile has just queued — the common call, once per tile and wave (93 % of the tiles
// of the headline workload queue something) — with everything that is the same for all of them kept scalar:
// the tile's buffer position is wave-uniform, and from the second tile of the buffer on no window is clamped
// at the start of the sequence, so window start = tile base + ((offset - s - k) & ~3) is one 32-bit lane value
// next to a scalar base (global_load_lds with an SGPR base).  ~9 VALU per call instead of ~40 (64-bit
// multiply-adds, compares and selects per lane): 262 000 calls per GiB.  `tile_local` = tile * FZ_TILE_BYTES >= one tile.
__device__ __forceinline__ void fz_prefetch_tile(const uint8_t *__restrict__ buf, const FzScanArgs &a, const FzWaveLds &w,
                                                 uint32_t qf, uint32_t qn, uint64_t tile_local) {
    const uint32_t lane = fz_lane();
    asm volatile("" ::: "memory");                         // the queue stores of this wave are issued before the reads below
    constexpr uint32_t BIAS = 2048u;                       // > FZ_MAX_M + FZ_MAX_K: keeps the lane offset non-negative
    const uint8_t *sbase = buf + fz_bcast64(tile_local) - BIAS;
    const uint32_t c0 = a.g0 * a.L + a.k;
    for (uint32_t e0 = qf; e0 < qn; e0 += 64u) {
        const uint32_t e = e0 + lane;
        if (e < qn) {
            const uint32_t code = w.queue[e];
            const uint32_t reach = ((code >> FZ_TILE_BITS) & (FZ_MAX_BLOCKS_PER_LAUNCH - 1u)) * a.L + c0;           // (g0 + block) * L + k
            const uint32_t voff = (((code & (FZ_TILE_BYTES - 1u)) - reach) & ~3u) + BIAS;
            for (uint32_t c = 0; c < a.win_pieces; ++c) {
                const uint32_t lds_dst = fz_uniform(w.win_lds + (c * a.qcap + e0) * 16u);
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff + 16u * c), "s"(sbase), "s"(lds_dst) : "memory");
            }
        }
    }
}

// Process queue entries [0, qn): range-check, then verify in place (FUSED) or confirm against HBM and
// bulk-append to the global hit list.  SEG: the sequence is a batch of file chunks (segments), a
// position may belong to two of them and is range-checked / verified once per segment; compiled
// separately because the in-memory search (one segment, the whole sequence) must not pay for it in
// registers.  FUSED && !SEG: the windows were prefetched (entry e -> slot e).  Returns the number of
// confirmed n-gram hits.
// Generic search, window table (fz_device.h: FzGenDedup): the scan enters every n-gram hit it lists — its window's slot
// (claimed by atomicCAS), the hit as a member of the window, and the window's leader = the hit of the smallest block
// (atomicMax on the inverted word) — so that when the automaton kernel starts, every hit knows whether it runs.
__device__ __forceinline__ void fz_gen_claim(const FzScanArgs &a, uint64_t hit, unsigned long long q) {
    if (q >= FZ_GEN_ORDER_MAX) return;
    const FzGenDedup dd(a.gen_dedup);
    const uint32_t blk = fz_hit_block(hit);
    const unsigned long long wk = fz_hit_index(hit) + a.k - (unsigned long long)blk * a.L + 1ull;
    uint32_t slot = (uint32_t)((wk * 0x9E3779B97F4A7C15ull) >> 40) & (FZ_GEN_DEDUP_SLOTS - 1u);
    uint32_t at = FZ_GEN_DEDUP_NONE;
    for (uint32_t probe = 0; probe < 32u; ++probe) {
        const unsigned long long old = atomicCAS(&dd.keys[slot], 0ull, wk);
        if (old == 0ull || old == wk) { at = slot; break; }
        slot = (slot + 1u) & (FZ_GEN_DEDUP_SLOTS - 1u);
    }
    if (at != FZ_GEN_DEDUP_NONE) {                            // (a crowded table or a full member list: the hit stays on its own)
        const uint32_t pos = atomicAdd(&dd.nmem[at], 1u);
        if (pos < FZ_GEN_DEDUP_MEMBERS) dd.mem[at * FZ_GEN_DEDUP_MEMBERS + pos] = (uint32_t)q;
        else at = FZ_GEN_DEDUP_NONE;
    }
    if (at != FZ_GEN_DEDUP_NONE) atomicMax(&dd.best[at], ~(((unsigned long long)blk << 32) | q));
    dd.wslot[q] = at;
}

template <int GW>
__device__ __forceinline__ uint32_t fz_flush_wf(const uint8_t *__restrict__ buf, const FzScanArgs &a, const uint8_t *lds0,
                                                const uint8_t *pat_lds, const FzWaveLds &w, const uint8_t *area, uint32_t per_wave,
                                                volatile uint32_t *fills, uint32_t wave, uint32_t qn, bool pooled,
                                                FzRec *__restrict__ recs, unsigned long long *__restrict__ counters);

template <bool FUSED, bool SEG>
__device__ __forceinline__ uint32_t fz_queue_flush(const uint8_t *__restrict__ buf, const FzScanArgs &a,
                                                   const uint8_t *pat_lds, const FzWaveLds &w, uint32_t qn,
                                                   uint64_t *__restrict__ hits, FzRec *__restrict__ recs,
                                                   unsigned long long *__restrict__ counters) {
    constexpr bool PREF = FUSED && !SEG;
    const uint32_t lane = fz_lane();
    uint32_t confirmed = 0;
    if (PREF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every prefetched window has landed in LDS
    fz_wave_lds_sync();
    const uint32_t width = (FUSED && !PREF) ? a.vlanes : 64u;  // candidates handled per pass
    const uint32_t ncand = (SEG && FUSED) ? fz_segment_candidates(a.geom) : 1u;
    for (uint32_t e0 = 0; e0 < qn; e0 += width) {
        for (uint32_t c = 0; c < ncand; ++c) {               // one inlined copy of the verification for both segments
            const uint32_t e = e0 + lane;
            bool valid = lane < width && e < qn;
            uint64_t hit = 0;
            uint64_t local = 0;
            uint32_t blk = 0;
            FzSeg sg;
            sg.sa = 0; sg.se = a.geom.n; sg.j = 0; sg.ok = 1;
            if (valid) {
                local = fz_code_local(w.queue[e], blk);
                const uint64_t idx = a.geom.buf_off + local;
                if (!SEG) {
                    valid = fz_hit_in_range(a, blk, idx, sg);
                } else if (FUSED) {
                    sg = fz_segment(a.geom, idx, c);
                    valid = fz_hit_in_range(a, blk, idx, sg);
                } else {                                      // hit list: accepted by ANY segment (the verify kernel re-checks)
                    valid = false;
                    for (uint32_t cc = 0; cc < fz_segment_candidates(a.geom); ++cc)
                        valid = valid || fz_hit_in_range(a, blk, idx, fz_segment(a.geom, idx, cc));
                }
                hit = fz_hit_pack(a.g0 + blk, idx);
            }
            if (FUSED) {
                if (SEG && c && !__ballot(valid)) continue;
                confirmed += fz_wave_verify<4, PREF>(buf, a, pat_lds, w, lane, hit, sg, valid, recs, counters,
                                                     PREF ? reinterpret_cast<const uint8_t *>(w.win) + e * 16u : nullptr);
            } else {
                if (valid) valid = fz_confirm(buf, a, blk, local);
                const unsigned long long mask = __ballot(valid);
                if (mask) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(&counters[0], (unsigned long long)__popcll(mask));
                    base = fz_bcast64(base);
                    const unsigned long long slot = base + fz_rank(mask);
                    if (valid && slot < a.hit_cap) {
                        hits[slot] = hit;
                        if (!SEG && a.gen_dedup) fz_gen_claim(a, hit, slot);
                    }
                }
            }
        }
    }
    fz_wave_lds_sync();
    return confirmed;
}

// The LAST flush of a workgroup of the fused in-memory scan, pooled: when all tiles are done every wave holds a
// partly filled queue (on the headline workload ~32 entries: verifying them per wave keeps half the lanes idle,
// and the scan is short of VALU issue slots, not of latency).  The four waves publish their fills, meet at one
// barrier and verify the concatenation of the four queues 64 entries at a time, pass p on wave p mod 4: a lane
// finds the wave that owns its entry by three compares and reads code and prefetched window out of that wave's
// area.  `area` = first wave's queue, `per_wave` = bytes per wave area, `fills` = four LDS dwords.
template <int MAXK, int BITS = 0>
__device__ __forceinline__ uint32_t fz_pooled_flush(const uint8_t *__restrict__ buf, const FzScanArgs &a,
                                                    const uint8_t *pat_lds, const uint8_t *area, uint32_t per_wave,
                                                    volatile uint32_t *fills, uint32_t wave, uint32_t qn,
                                                    FzRec *__restrict__ recs, unsigned long long *__restrict__ counters,
                                                    const uint8_t *peq_tabs = nullptr) {
    const uint32_t lane = fz_lane();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's prefetched windows have landed in LDS
    if (lane == 0) fills[wave] = qn;
    __syncthreads();
    const uint32_t n0 = fills[0], n1 = fills[1], n2 = fills[2], n3 = fills[3];
    const uint32_t total = n0 + n1 + n2 + n3;
    FzWaveLds none;
    none.queue = nullptr; none.win = nullptr; none.scores = nullptr; none.win_lds = 0;
    uint32_t confirmed = 0;
    for (uint32_t e0 = wave * 64u; e0 < total; e0 += FZ_WAVES_PER_BLOCK * 64u) {
        uint32_t li = e0 + lane, ow = 0;
        bool valid = li < total;
        if (li >= n0) { li -= n0; ow = 1; if (li >= n1) { li -= n1; ow = 2; if (li >= n2) { li -= n2; ow = 3; } } }
        const uint8_t *mine = area + ow * per_wave;
        uint64_t hit = 0;
        FzSeg sg;
        sg.sa = 0; sg.se = a.geom.n; sg.j = 0; sg.ok = 1;
        if (valid) {
            uint32_t blk;
            const uint64_t idx = a.geom.buf_off + fz_code_local(reinterpret_cast<const uint32_t *>(mine)[li], blk);
            valid = fz_hit_in_range(a, blk, idx, sg);
            hit = fz_hit_pack(a.g0 + blk, idx);
        }
        confirmed += fz_wave_verify<MAXK, true, BITS>(buf, a, pat_lds, none, lane, hit, sg, valid, recs, counters,
                                                      mine + a.qcap * 4u + li * 16u, peq_tabs);
    }
    return confirmed;
}

// Mid-scan flush of the fused bit-vector forms (in-memory Levenshtein searches, fz_scan_kernel<..., WFG = 1 / 2>; NW = 0:
// the register band / Hamming count under the same queue discipline, WFG = 3).  A pass
// costs the wave ~m - L + 2k columns whatever its number of candidates, so passes are FULL: the queue is worked off 64
// entries at a time, and what is left below 64 moves to the front of the queue — code and prefetched window pieces —
// and waits for the next tiles' entries (only a queue that holds fewer than 64 in the first place is verified as it is:
// the scan loop asks for that when it expects the next tile to overflow it).  `qn` -> the entries kept.
template <int NW>
__device__ __forceinline__ uint32_t fz_bits_flush(const uint8_t *__restrict__ buf, const FzScanArgs &a, const uint8_t *pat_lds,
                                                  const uint8_t *peq_tabs, const FzWaveLds &w, uint32_t &qn,
                                                  FzRec *__restrict__ recs, unsigned long long *__restrict__ counters) {
    const uint32_t lane = fz_lane();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every prefetched window has landed in LDS
    fz_wave_lds_sync();
    const uint32_t end = qn >= 64u ? qn & ~63u : qn;
    uint32_t confirmed = 0;
    for (uint32_t e0 = 0; e0 < end; e0 += 64u) {
        const uint32_t e = e0 + lane;
        bool valid = e < end;
        uint64_t hit = 0;
        FzSeg sg;
        sg.sa = 0; sg.se = a.geom.n; sg.j = 0; sg.ok = 1;
        if (valid) {
            uint32_t blk;
            const uint64_t idx = a.geom.buf_off + fz_code_local(w.queue[e], blk);
            valid = fz_hit_in_range(a, blk, idx, sg);
            hit = fz_hit_pack(a.g0 + blk, idx);
        }
        confirmed += fz_wave_verify<4, true, NW>(buf, a, pat_lds, w, lane, hit, sg, valid, recs, counters,
                                                 reinterpret_cast<const uint8_t *>(w.win) + e * 16u, peq_tabs);
    }
    const uint32_t rem = qn - end;                          // < 64, and only behind at least one full pass
    if (lane < rem) {
        w.queue[lane] = w.queue[end + lane];
        uint8_t *wb = reinterpret_cast<uint8_t *>(w.win);
        for (uint32_t c = 0; c < a.win_pieces; ++c)
            *reinterpret_cast<uint4 *>(wb + (c * a.qcap + lane) * 16u) = *reinterpret_cast<const uint4 *>(wb + (c * a.qcap + end + lane) * 16u);
    }
    qn = rem;
    fz_wave_lds_sync();
    return confirmed;
}

// NWIN  : 1 -> hash = (masked dword at the offset) * K               (L <= 4, v_mul_lo_u32);
//         2 -> hash = low24(dword at offset + DH) * K + dword at offset (v_mad_u32_u24), DH = min(L, 8) - 3.
// FUSED : verify candidates inside this kernel (records out) or emit exact hits (hit list out).
// SEG   : the buffer is a batch of file chunks with per-chunk clamps (find_near_matches_in_file).
// SA    : the table slot comes from hash bits 2..6 as they are (one v_and per offset instead of v_lshrrev + v_and;
//         measured -7 % on the headline workload: the fused scan is bound by VALU issue, +/- one full-rate VALU op
//         per offset = +10 / -7 %).  Those bits only depend on window bytes 0 and DH, so the host can use this form
//         when the blocks of a launch differ there (fzhip.hip: choose_launch_blocks); the general form takes any
//         five hash bits.
// Each thread owns 16 consecutive byte offsets per row and reads 24 bytes (16 + 8 halo).
// Block test: slot = (hash >> lut_shift) & 31; lut[slot] holds the hash of the block that lives there
// (the host picks K and lut_shift so that different block hashes get different slots) or, for a free
// slot, a value that belongs to another slot, so hash ^ lut[slot] == 0 <=> the window hashes like
// some block.  32 slots of 4 bytes = one slot per LDS bank: lanes that read different slots never
// collide, lanes that read the same slot are served by one broadcast.
// Measured against per-block VALU compares (benchmarks/filter_variants.hip, v3 vs v14, 3 blocks,
// L2-resident data): 0.222 -> 0.175 ms per GiB, and no longer growing with the number of blocks.
// Fast hits are queued per wave ACROSS tiles and processed 64 at a time (full lanes, one latency
// chain per ~100 candidates instead of one per tile).  A tile denser than the queue is re-scanned
// by enumeration ("slow tile": correctness path for pathological inputs).
// Half-tile software pipeline: rows 0-1 of the NEXT tile are requested before rows 2-3 of this one are
// tested, so a wave always has two rows (3 KiB) of loads in flight while it computes, on the same 24 data
// VGPRs as "load a tile, test a tile" (measured: 0.236 -> 0.227 ms on the headline workload).
// What round 2 measured about this kernel (benchmarks/lab_*.sh, 1 GiB DNA, |p| = 20, k = 2):
//   * one VALU op less per offset (slot address by v_and only): no change -> the hot loop is not bound by
//     VALU issue; the scan without candidates takes 0.200 ms at 7, 6, 5 and 4 workgroups per CU alike;
//   * without verification 0.206 ms, with the window fetch but no DP 0.225 ms, complete 0.226 ms: what
//     candidates cost was the exposed fetch of their windows at the end of every wave's life -> prefetched
//     by LDS-DMA now (fz_prefetch_windows).
// 7 waves per SIMD (72 VGPRs): measured 2-3 % faster than the natural 79-VGPR / 6-wave allocation;
// 8 waves (64 VGPRs) spills 27 VGPRs in the verify path and is 50 % slower.
template <int NWIN, int DH, bool FUSED, bool SEG, bool SA, int WFG = 0>
// (the bit-vector forms: 6 waves per SIMD = 80 VGPRs with one-word columns, 5 = 96 with two-word ones — their queues and Peq
//  tables leave LDS for at most that many workgroups per CU anyway, and the column loop keeps the next column's Peq word and
//  character in flight)
// (round 6, the equal-n-gram sets in the rare path: the file API's fused instances and the lane-per-cell forms, which sat at
//  exactly 72 VGPRs, take 6 waves as well — the former are bound by the host's copies, the latter serve patterns beyond 128
//  characters only)
#define FZ_SCAN_WAVES(WFG, FUSED, SEG) ((WFG) == 2 ? 5 : ((WFG) != 0 || ((FUSED) && (SEG))) ? 6 : 7)      // (WFG = 4: 6 as well)
__global__ __launch_bounds__(FZ_FILTER_THREADS)
__attribute__((amdgpu_waves_per_eu(FZ_SCAN_WAVES(WFG, FUSED, SEG), FZ_SCAN_WAVES(WFG, FUSED, SEG)))) void fz_scan_kernel(
    const uint8_t *__restrict__ buf, const FzScanArgs a, uint64_t ntiles,
    uint64_t *__restrict__ hits, FzRec *__restrict__ recs, unsigned long long *__restrict__ counters) {
    constexpr bool WF = WFG == 16 || WFG == 32;       // lane-per-cell verification inside the scan, WFG lanes per candidate
    constexpr int BITS = (WFG == 1 || WFG == 2 || WFG == 4) ? WFG : 0;   // bit-vector verification inside the scan, one candidate per lane:
                                                                         // one / two 64-bit words per column, 4 = one 32-bit word
    // WFG = 3: the Hamming count of WFG = 0 (substitutions-only searches) under the queue discipline of the bit-vector forms
    // (full passes, block-range passes over dense tiles), for patterns that let expect dense candidates
    constexpr bool ADAPT = BITS != 0 || WFG == 3;
    constexpr int VF = WFG == 3 ? -1 : BITS;              // what fz_wave_verify runs: -1 Hamming count only, 0 by mode, 1 / 2 bit vectors
    static_assert(WFG == 0 || ADAPT || WF, "0: register band / Hamming count; 3: Hamming count; 1, 2, 4: bit-vector columns; 16, 32: lanes per candidate");
    static_assert(WFG == 0 || (FUSED && !SEG), "the lane-per-cell and bit-vector forms are fused forms of the in-memory search");
    constexpr bool PREF = FUSED && !SEG && !WF;       // candidate windows are prefetched by LDS-DMA
    constexpr uint32_t peq_bytes = BITS ? FZ_PEQ_BYTES(BITS ? BITS : 1) : 0u;   // the two Peq tables behind the pattern
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t mpad = FUSED ? (a.m + 15u) & ~15u : 0u;   // only the fused verification reads the pattern from LDS (m <= FZ_MAX_M there)
    // [32] hash living in the slot.  The kernel has no static LDS, so the dynamic area, and with it this
    // table, starts at LDS address 0 and a slot's byte offset is its address (saves one VALU add per
    // lookup); trap if a toolchain ever lays LDS out differently.
    uint32_t *lut = reinterpret_cast<uint32_t *>(smem);
    if (reinterpret_cast<uintptr_t>((FzLdsU8 *)smem) != 0) __builtin_trap();
    uint8_t *pat_lds = smem + FZ_TABLE_BYTES;
    if constexpr (FUSED)
        for (uint32_t i = threadIdx.x; i < a.m; i += FZ_FILTER_THREADS) pat_lds[i] = a.pat[i];
    if constexpr (BITS != 0) {
        // Peq tables (fz_device.h: fz_verify_lev_bits): zero, then one LDS atomic per pattern position and table
        constexpr int NWc = BITS ? BITS : 1;
        uint32_t *peq = reinterpret_cast<uint32_t *>(smem + FZ_TABLE_BYTES + mpad);
        constexpr uint32_t wdw = FZ_BITS_WIDTH(NWc) / 32u;                   // dwords per table word
        for (uint32_t i = threadIdx.x; i < peq_bytes / 4u; i += FZ_FILTER_THREADS) peq[i] = 0u;
        __syncthreads();
        if (threadIdx.x < a.m) {
            const uint32_t q = threadIdx.x, c = pat_lds[q];
            const uint32_t bf = fz_bits_fwd_bit<NWc>(a.m, q), br = fz_bits_rev_bit<NWc>(a.m, q);
            atomicOr(&peq[c * wdw + (bf >> 5)], 1u << (bf & 31u));
            atomicOr(&peq[(256u + c) * wdw + (br >> 5)], 1u << (br & 31u));
        }
    }
    if (threadIdx.x < FZ_LUT_SLOTS) {
        uint32_t t = ((threadIdx.x + 1u) & (FZ_LUT_SLOTS - 1u)) << a.lut_shift;   // free slot: a value of the next slot
        uint32_t who = 0xffu;                                                     // ... and the block that lives in the slot
        uint32_t set = 0;                                                         // ... or, with equal n-grams in the launch, all of them
        for (uint32_t g = a.nblk; g-- > 0;)
            if (((a.H[g] >> a.lut_shift) & (FZ_LUT_SLOTS - 1u)) == threadIdx.x) { t = a.H[g]; who = g; set |= 0x10000u << g; }
        lut[threadIdx.x] = t;
        lut[FZ_LUT_SLOTS + threadIdx.x] = (a.flags & FZ_FLAG_DUP_HASHES) ? (who | set) : who;
    }
    const bool dup_hashes = (a.flags & FZ_FLAG_DUP_HASHES) != 0;
    // this workgroup's walk over the tiles: first_tile, first_tile + stride, .. below limit (scalar values); the flushes
    // decode queue entries with the copy in LDS (fz_code_local)
    uint32_t wg0 = 0, stride = gridDim.x;
    uint64_t tile0 = 0, limit = ntiles;
#pragma unroll
    for (uint32_t r = 0; r < FZ_MAX_REGIONS; ++r)
        if (r < a.nreg && blockIdx.x >= a.reg_wg0[r]) { wg0 = a.reg_wg0[r]; stride = a.reg_nwg[r]; tile0 = a.reg_tile0[r]; limit = a.reg_end[r]; }
    const uint64_t first_tile = tile0 + (blockIdx.x - wg0);
    if (threadIdx.x == 0) {
        uint32_t *walk = reinterpret_cast<uint32_t *>(smem + FZ_WALK_LDS);
        walk[0] = (uint32_t)first_tile; walk[1] = (uint32_t)(first_tile >> 32); walk[2] = stride;
    }
    __syncthreads();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t qcap = PREF ? a.qcap : (uint32_t)FZ_QCAP;   // queue entries per wave
    const FzWaveLds w = PREF ? fz_wave_lds_pref(smem + FZ_TABLE_BYTES + mpad + peq_bytes, FZ_TABLE_BYTES + mpad + peq_bytes, wave, qcap, a.win_pieces)
                        : WF ? fz_wave_lds(smem + FZ_TABLE_BYTES + mpad, wave, fz_wf_fused_dwords(a.win_dwords, WF ? WFG : 16), 0u, 1u, true)
                             : fz_wave_lds(smem + FZ_TABLE_BYTES + mpad, wave, FUSED ? a.win_dwords : 0u,
                                           FUSED ? a.band_w : 0u, a.vlanes, true);
    const uint32_t hash_k = a.hash_k;
    // byte address of a hash's slot = (h >> (lut_shift - 2)) & 0x7c: two VGPR-only VALU ops (a shift
    // amount in an SGPR or an SDWA byte select would issue at half the rate, benchmarks/valu_rates.hip)
    uint32_t slot_shift;
    asm volatile("v_mov_b32 %0, %1" : "=v"(slot_shift) : "s"(a.lut_shift - 2u));
    const uint32_t mask1 = a.mask1;
    const uint32_t lane = fz_lane();
    const uint32_t lane_off = threadIdx.x * 16u;
    uint32_t qn = 0;                                  // wave-uniform queue fill
    uint32_t qf = 0;                                  // PREF: entries [0, qf) have their windows requested
    uint32_t confirmed = 0;                           // wave-uniform statistics
    uint32_t titer = 0;                               // tile iteration of this workgroup
    uint64_t tile = first_tile;
    // has_near_match_* (substitutions_only.py:218-233 stops at the first match): a workgroup that starts after a record
    // has been counted skips its tiles (thousands of short workgroups per launch: the ones not yet started are the saving)
    if ((a.flags & FZ_FLAG_ANY) && __hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) tile = limit;
    bool slow = false;                                // a tile is being re-scanned by enumeration
    uint32_t slow_pos = 0;
    // Bit-vector form: the queue is worked off in full passes (fz_bits_flush) and filled as far as the tiles' recent yield
    // lets expect it to hold: `ylast` = entries the last tile queued (wave-uniform).  The expectation only steers; a tile that
    // overflows the queue all the same is scanned again behind a flush (and by enumeration if it overflows an empty queue).
    uint32_t ylast = 0;
    // ... and where the data is denser than any queue (DNA with 4-character n-grams: hundreds of hits per tile and wave), a
    // tile is taken in several passes, each for the blocks [b0, b0 + bw) of the launch only: a tile that overflows the EMPTY
    // queue halves bw and starts again, tiles that queue little double it.  Only a tile that overflows the empty queue with
    // one block (a run of one character meeting an n-gram of that character) is enumerated.
    uint32_t b0 = 0, bw = FZ_MAX_BLOCKS_PER_LAUNCH;

    // the filter over one row (row R of the tile)
    auto test_row = [&](const uint4 &v, const uint2 &h, auto Rc) {
        constexpr int r = decltype(Rc)::value;
        // byte offsets tested per wave-uniform branch: 8 in the hit-emitting form when the n-grams have 8 bytes or
        // more (DH == 5) — such n-grams are rare in any data, the branch is hardly ever taken and one compare serves
        // twice the offsets (exact search of a 20-byte pattern: 0.199 -> 0.194 ms per GiB); 4 otherwise (on DNA with
        // 6-byte n-grams 17 % of the 4-offset groups fire: with 8 the rare path's compares double, 0.223 -> 0.246 ms;
        // the fused form has no registers to spare for 8 hashes: 26 VGPRs spilled)
        constexpr int GRP = FZ_GROUP ? FZ_GROUP : (NWIN == 2 && DH == 5 && !FUSED ? 8 : 4);
        const uint32_t w6[6] = {v.x, v.y, v.z, v.w, h.x, h.y};
#pragma unroll
        for (int j = 0; j < 16 / GRP; ++j) {     // GRP byte offsets per ballot
            uint32_t hv[GRP], lv[GRP];                    // window hashes and the hashes living in their table slots
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int o = GRP * j + i;
                const uint32_t x = FZ_WIN(w6, o);
                if (NWIN == 1) hv[i] = (x & mask1) * hash_k;                             // v_mul_lo_u32
                else hv[i] = __umul24(FZ_WIN(w6, o + DH), hash_k) + x;                   // v_mad_u32_u24
                uint32_t slot4;
                if constexpr (SA) asm("v_and_b32 %0, " FZ_LUT_ADDR_MASK_STR ", %1" : "=v"(slot4) : "v"(hv[i]));
                else asm("v_lshrrev_b32 %0, %1, %2\n\tv_and_b32 %0, " FZ_LUT_ADDR_MASK_STR ", %0" : "=v"(slot4) : "v"(slot_shift), "v"(hv[i]));
                lv[i] = *reinterpret_cast<FzLdsU32 *>(slot4);            // lut sits at LDS address 0
            }
            // (measured and not kept: one v_cmp per offset with the lane masks OR-ed on the scalar unit instead of
            //  xor / min3 / min / one v_cmp per four offsets — 15.5 instead of 22.5 VALU per group, 0.2222 vs 0.2195 ms)
            uint32_t am[GRP];
#pragma unroll
            for (int i = 0; i < GRP; ++i) am[i] = hv[i] ^ lv[i];
            uint32_t acc = min(min(am[0], am[1]), min(am[2], am[3]));
            if constexpr (GRP == 8) acc = min(acc, min(min(am[4], am[5]), min(am[6], am[7])));
            const bool fire = __ballot(acc == 0) != 0;
            if (__builtin_expect(fire, 0)) {              // wave-uniform, rare: some lane, some offset
#pragma unroll
                for (int i = 0; i < GRP; ++i) {
                    const unsigned long long mi = __ballot(hv[i] == lv[i]);      // which offset (scalar branch)
                    if (mi) {
                        // which block: the window's hash equals the one in its slot, and the dword behind the hash
                        // table says whose that is (one LDS read instead of a compare per block)
                        uint32_t slot4;
                        if constexpr (SA) asm("v_and_b32 %0, " FZ_LUT_ADDR_MASK_STR ", %1" : "=v"(slot4) : "v"(hv[i]));
                        else asm("v_lshrrev_b32 %0, %1, %2\n\tv_and_b32 %0, " FZ_LUT_ADDR_MASK_STR ", %0" : "=v"(slot4) : "v"(slot_shift), "v"(hv[i]));
                        const uint32_t g = *reinterpret_cast<FzLdsU32 *>(slot4 + FZ_LUT_BYTES);
                        // the queue code is recomputed here: a (tid << 4 | titer << 18) kept in a VGPR across the tile
                        // saves three ops per firing but is the register that spills (measured: 0.218 -> 0.221 ms)
                        uint32_t pos = threadIdx.x;
                        asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(pos));
                        if (__builtin_expect(dup_hashes, 0)) {
                            // equal n-grams (equal hashes) share a slot: the dword behind the hash table then carries, from
                            // bit 16 up, the SET of the launch's blocks that live in the slot, and a firing lane queues one
                            // entry per member (rounds 1 - 5 compared the window's hash with every block of the launch, offset
                            // by offset: ~8 scalar instructions per block and offset of a fired group — a DNA pattern with a
                            // repeated 4-character n-gram ran 4 x slower than one without)
                            uint32_t set = hv[i] == lv[i] ? g >> 16 : 0u;
                            pos += (uint32_t)(r * FZ_ROW_BYTES + GRP * j + i);
                            while (__ballot(set != 0u)) {
                                const uint32_t gb = (uint32_t)__ffs((int)set) - 1u;       // (an empty set: 0xffffffff, never taken)
                                const bool take = set != 0u && (!ADAPT || gb - b0 < bw);
                                const unsigned long long mk = __ballot(take);
                                const uint32_t slot = qn + fz_rank(mk);
                                if (take && slot < qcap) w.queue[slot] = fz_code(pos, gb, titer);
                                qn += (uint32_t)__popcll(mk);
                                set &= set - 1u;
                            }
                        } else if constexpr (ADAPT) {
                            // only the blocks of this pass over the tile (b0, bw below): g = 0xff (a free slot) never passes
                            const bool take = hv[i] == lv[i] && g - b0 < bw;
                            const unsigned long long mt = __ballot(take);
                            const uint32_t slot = qn + fz_rank(mt);
                            if (take && slot < qcap)
                                w.queue[slot] = fz_code(pos + (uint32_t)(r * FZ_ROW_BYTES + GRP * j + i), g, titer);
                            qn += (uint32_t)__popcll(mt);
                        } else {
                            const uint32_t slot = qn + fz_rank(mi);
                            if (hv[i] == lv[i] && slot < qcap)
                                w.queue[slot] = fz_code(pos + (uint32_t)(r * FZ_ROW_BYTES + GRP * j + i), g, titer);
                            qn += (uint32_t)__popcll(mi);
                        }
                    }
                }
            }
        }
    };

    for (;;) {
        if (slow) {
            // enumerate (row, offset, block) candidates of tile `tile`, 64 lanes at a time
            const uint32_t nb = ADAPT ? min(bw, a.nblk - b0) : a.nblk;      // (bit-vector form: the blocks of this pass)
            const uint32_t steps = FZ_FILTER_ROWS * 16u * nb;
            while (slow_pos < steps && qn + 64u <= qcap) {
                const uint32_t blk = (ADAPT ? b0 : 0u) + slow_pos % nb;
                const uint32_t ro = slow_pos / nb;
                w.queue[qn + lane] = fz_code((ro >> 4) * FZ_ROW_BYTES + lane_off + (ro & 15u), blk, titer);
                qn += 64u;
                ++slow_pos;
            }
            if (slow_pos >= steps) {
                slow = false;
                if (ADAPT && b0 + bw < a.nblk) b0 += bw;
                else { b0 = 0; tile += stride; ++titer; }
            }
        } else if (tile < limit && (ADAPT ? (qn == 0u || qn + ylast + (ylast >> 2) + 8u <= qcap) : qn <= qcap / 2)) {
            uint4 va[2], vb[2];
            uint2 ha[2], hb[2];
            bool pre;                                 // va / ha hold rows 0-1 of the next tile
            {
                const uint8_t *tsrc = buf + fz_bcast64(tile * (uint64_t)FZ_TILE_BYTES);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    va[r] = *reinterpret_cast<const uint4 *>(tsrc + r * FZ_ROW_BYTES + lane_off);
                    ha[r] = *reinterpret_cast<const uint2 *>(tsrc + r * FZ_ROW_BYTES + lane_off + 16);
                }
            }
            do {
                const uint8_t *tsrc = buf + fz_bcast64(tile * (uint64_t)FZ_TILE_BYTES);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    vb[r] = *reinterpret_cast<const uint4 *>(tsrc + (r + 2) * FZ_ROW_BYTES + lane_off);
                    hb[r] = *reinterpret_cast<const uint2 *>(tsrc + (r + 2) * FZ_ROW_BYTES + lane_off + 16);
                }
                __builtin_amdgcn_sched_barrier(0);    // all loads are issued before the first use
                const uint32_t q_tile = qn;
                test_row(va[0], ha[0], std::integral_constant<int, 0>{});
                test_row(va[1], ha[1], std::integral_constant<int, 1>{});
                const bool same_tile = ADAPT && b0 + bw < a.nblk;       // the next pass is over this tile again (its other blocks)
                const uint64_t next = same_tile ? tile : tile + stride;
                if constexpr (ADAPT) pre = next < limit && qn + 3u * (qn - q_tile) + 8u <= qcap;   // this pass's second half + the next pass
                else pre = next < limit && qn <= qcap / 2;
                {   // unconditional (a branch here would make the compiler wait for the prefetch at the join):
                    // without a next tile the loads re-read this one (L2 hits, results unused)
                    const uint8_t *nsrc = buf + fz_bcast64((pre ? next : tile) * (uint64_t)FZ_TILE_BYTES);
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        va[r] = *reinterpret_cast<const uint4 *>(nsrc + r * FZ_ROW_BYTES + lane_off);
                        ha[r] = *reinterpret_cast<const uint2 *>(nsrc + r * FZ_ROW_BYTES + lane_off + 16);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                test_row(vb[0], hb[0], std::integral_constant<int, 2>{});
                test_row(vb[1], hb[1], std::integral_constant<int, 3>{});
                if (qn > qcap) {                      // this tile overflowed the queue: drop its
                    qn = q_tile;                      // partial entries and re-scan it by enumeration
                    if constexpr (ADAPT) {        // ... or, bit-vector form:
                        if (q_tile != 0u) {           // once more behind a flush of what the queue held,
                            ylast = qcap;
                            break;
                        }
                        const uint32_t nb = min(bw, a.nblk - b0);
                        if (nb > 1u) {                // once more for half of the blocks (the queue is empty: no flush)
                            bw = (nb + 1u) >> 1;
                            ylast = qcap >> 1;
                            break;
                        }
                    }
                    slow = true;
                    slow_pos = 0;
                    break;
                }
                if constexpr (ADAPT) ylast = qn - q_tile;
                if (PREF && qn > qf) {
                    if (tile) fz_prefetch_tile(buf, a, w, qf, qn, tile * (uint64_t)FZ_TILE_BYTES);
                    else fz_prefetch_windows(buf, a, w, qf, qn);          // the first tile: windows clamped at the start
                    qf = qn;
                }
                if constexpr (ADAPT) {
                    if (same_tile) {
                        b0 += bw;
                    } else {
                        b0 = 0;
                        if (bw < a.nblk && ylast <= (qcap >> 3)) bw <<= 1;   // little queued: twice the blocks per pass from the next tile on
                        tile = next;
                        ++titer;
                    }
                } else {
                    tile = next;
                    ++titer;
                }
            } while (pre);                            // else: the end of the sequence, or a flush is due
        }
        const bool done = !slow && tile >= limit;
        if (PREF && done) break;                      // what is queued now is verified by the pooled flush below
        if constexpr (WF) {
            // lane-per-cell verification (Levenshtein budgets 5 .. 15): own queue in mid-scan, the workgroup's pool at the end
            confirmed += fz_flush_wf<WF ? WFG : 16>(buf, a, smem, pat_lds, w, smem + FZ_TABLE_BYTES + mpad,
                                                    fz_wave_lds_bytes(fz_wf_fused_dwords(a.win_dwords, WF ? WFG : 16), 0u, 1u, true),
                                                    reinterpret_cast<volatile uint32_t *>(smem + 2u * FZ_LUT_BYTES), wave, qn, done, recs, counters);
        } else if (qn) {
            if (PREF && qn > qf) fz_prefetch_windows(buf, a, w, qf, qn);
            if constexpr (ADAPT) {
                confirmed += fz_bits_flush<VF>(buf, a, pat_lds, smem + FZ_TABLE_BYTES + mpad, w, qn, recs, counters);
                qf = qn;                              // what stays queued has its window
                continue;
            } else {
                confirmed += fz_queue_flush<FUSED, SEG>(buf, a, pat_lds, w, qn, hits, recs, counters);
            }
        }
        qn = 0;
        qf = 0;
        if (done) break;
    }
    if constexpr (PREF) {
        if (qn > qf) fz_prefetch_windows(buf, a, w, qf, qn);
        confirmed += fz_pooled_flush<4, VF>(buf, a, pat_lds, smem + FZ_TABLE_BYTES + mpad + peq_bytes, fz_wave_lds_pref_bytes(qcap, a.win_pieces),
                                              reinterpret_cast<volatile uint32_t *>(smem + 2u * FZ_LUT_BYTES), wave, qn, recs, counters,
                                              smem + FZ_TABLE_BYTES + mpad);
    }

    // (measured and not kept: one no-return atomic per workgroup — ticket and tallies in one word — with the last-indexed
    // workgroup polling for the others instead of every workgroup waiting for its ticket: 0.2172 vs 0.2183 ms, within noise)
    if (FUSED && lane == 0 && confirmed) atomicAdd(&counters[8 + (blockIdx.x & 63u)], (unsigned long long)confirmed);
    fz_finish_launch(a, counters, lut);
}

// Verification of a hit list in HBM, one lane per candidate (parameter ranges whose LDS footprint does
// not fit beside the filter, substitutions with large budgets, Levenshtein budgets above 31).  One wave
// verifies 64 hits at a time.  Dynamic LDS: pattern + per-wave window/score areas.
__global__ __launch_bounds__(256) void fz_verify_kernel(const uint8_t *__restrict__ buf, const FzScanArgs a,
                                 const uint64_t *__restrict__ hits, FzRec *__restrict__ recs,
                                 unsigned long long *__restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t mpad = (a.m + 15u) & ~15u;
    uint8_t *pat_lds = smem;
    fz_copy_pattern(pat_lds, a, threadIdx.x, blockDim.x);
    __syncthreads();
    const FzWaveLds w = fz_wave_lds(smem + mpad, threadIdx.x >> 6, a.win_dwords, a.band_w, a.vlanes, false);
    unsigned long long nh = counters[0];
    if (nh > a.hit_cap) nh = a.hit_cap;
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t ncand = fz_segment_candidates(a.geom);
    for (uint64_t q0 = wave * a.vlanes; q0 < nh; q0 += waves * a.vlanes) {
        const uint64_t q = q0 + fz_lane();
        const bool have = fz_lane() < a.vlanes && q < nh;
        const uint64_t hit = have ? hits[q] : 0;
        for (uint32_t c = 0; c < ncand; ++c) {
            const FzSeg sg = fz_segment(a.geom, fz_hit_index(hit), c);
            const bool valid = have && fz_hit_in_range_s(a, fz_hit_block(hit) * a.L, fz_hit_index(hit), sg);
            if (!__ballot(valid)) continue;
            fz_wave_verify<FZ_REG_BAND_MAX, false>(buf, a, pat_lds, w, fz_lane(), hit, sg, valid, recs, counters);
        }
    }
    fz_finish_launch(a, counters, reinterpret_cast<uint32_t *>(smem));
}

// ---------------------------------------------------------------------------------------------
// Lane-per-DP-cell verification (the north star's wavefront form), for budgets 5 .. 31 where one
// lane per candidate is a long serial chain.  GW lanes own one candidate; lane gl holds the band
// cell D[i][i + gl - K] of the current row i (K = max_l_dist).  Row i + 1 needs
//   the diagonal  D[i][i+d]      = the lane's own cell,
//   the upper     D[i][i+1+d]    = the cell of lane gl + 1          (one DPP row shift),
//   the left      D[i+1][i+d]    = the NEW cell of lane gl - 1:  v[gl] = min_{e <= gl} (a[e] + gl - e),
//                                  a prefix-min over the lanes of a[e] - e (log2 GW DPP steps).
// Same table as fz_expand_band / c_expand_*: band K >= budget holds every cell <= budget, bottom row
// scanned for the LAST arg-min from the column-0 baseline (pyx:33-34, :67-69).
template <int GW>
__device__ __forceinline__ uint32_t fz_xl_shl1(uint32_t v, uint32_t fill, uint32_t gl) {
    if constexpr (GW == 16) {
        return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x101, 0xf, 0xf, false);   // row_shl:1
    } else {
        // lane l <- lane l + 1 across the wave (wave_shl:1; lane 63 <- fill), then the last lane of every group <- fill
        const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false);
        return gl == (uint32_t)GW - 1u ? fill : o;
    }
}

template <int GW>
__device__ __forceinline__ uint32_t fz_xl_prefix_min(uint32_t v, uint32_t gl) {
    if constexpr (GW == 16) {
        // v = min(v, v of the lane n below) in one instruction each: a DPP source outside the row disables
        // the lane, and since destination and second source are the same register such a lane keeps v
        asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                     : "+v"(v));
    } else {
        // inside the 16-lane rows as above, then the last lane of row 0 / 2 into rows 1 / 3 (32-lane groups) and the last
        // lane of row 1 into rows 2 and 3 (the whole wave): lanes without a source keep their value
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
        if constexpr (GW == 64)
            v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    }
    return v;
}

// The rows of one bounded expansion for every group of the wave.  All arguments but gl are uniform inside a
// group.  sub(i) = lds[sub_addr + i * sub_step], win(j) = lds[win_addr + j * win_step] (bytes in LDS;
// reads one element outside either range land in this workgroup's LDS and only feed cells that are
// forced anyway).  -> the lane's cell of the bottom row, D[sublen][sublen + gl - K] (garbage above `budget`
// once every live group has run out of cells within its budget: nothing can pass then).
template <int GW>
__device__ __forceinline__ uint32_t fz_wf_rows(const uint8_t *lds, uint32_t gl, uint32_t K, int sub_addr, int sub_step,
                                               uint32_t sublen, int win_addr, int win_step, uint32_t winlen,
                                               uint32_t budget, bool valid) {
    constexpr uint32_t INF = 0x3fffu;
    const int d = (int)gl - (int)K;
    const bool act = valid && gl <= 2u * K;
    uint32_t jv = act ? (uint32_t)d : 0x7fff0000u;          // column j = i + d of this lane (row 0); "negative" = huge
    uint32_t cell = jv <= winlen ? jv : INF;                // row 0: D[0][j] = j
    if (!valid) { win_addr = sub_addr = 0; win_step = sub_step = 0; }
    int caddr = win_addr + win_step * d;                    // row i compares win(i + d - 1)
    if (!act) caddr = win_addr;
    int paddr = sub_addr;
    uint32_t chr = lds[caddr], pc = lds[paddr];
    const uint32_t bias = (uint32_t)GW - gl;
    for (uint32_t i = 1;; ++i) {
        const bool live = valid && i <= sublen;
        if (!__ballot(live)) break;
        caddr += win_step;
        paddr += sub_step;
        const uint32_t nchr = lds[caddr], npc = lds[paddr];   // next row's characters: their latency hides behind this row
        jv += 1u;
        const uint32_t up = fz_xl_shl1<GW>(cell, INF, gl);
        uint32_t v = min(cell + (chr != pc ? 1u : 0u), up + 1u);
        if (jv == 0u) v = i;                                  // column 0: D[i][0] = i
        const bool bad = jv > winlen;
        v = bad ? INF : v;
        v = fz_xl_prefix_min<GW>(v + bias, gl) - bias;
        v = bad ? INF : v;
        if (live) cell = v;
        chr = nchr;
        pc = npc;
        // row minima never decrease: once no live candidate has a cell within its budget, nothing can pass
        if ((i & 3u) == 0u && !__ballot(live && cell <= budget)) break;
    }
    return cell;
}

// Bottom row -> (best, last arg-min) over the columns 1 .. winlen from the column-0 baseline
// (pyx:33-34, :67-69); uniform inside the group.  winlen may be smaller than the one the rows ran with:
// a cell only depends on cells of smaller or equal columns.
template <int GW>
__device__ __forceinline__ bool fz_wf_pick(uint32_t cell, uint32_t gl, uint32_t K, uint32_t sublen, uint32_t winlen,
                                           uint32_t budget, bool valid, uint32_t &dist, uint32_t &consumed) {
    const int jb = (int)sublen + (int)gl - (int)K;
    const bool validj = valid && gl <= 2u * K && jb >= 1 && jb <= (int)winlen;
    uint32_t key = validj ? ((cell << 8) | (255u - gl)) : 0xffffffffu;     // min cell, then the LARGEST column
    key = fz_xl_prefix_min<GW>(key, gl);
    key = (uint32_t)__shfl((int)key, (int)((fz_lane() & ~((uint32_t)GW - 1u)) + (uint32_t)GW - 1u), 64);
    uint32_t best = sublen, arg = 0;
    if (key != 0xffffffffu && (key >> 8) <= sublen) {
        best = key >> 8;
        arg = (uint32_t)((int)sublen + (int)(255u - (key & 255u)) - (int)K);
    }
    dist = best;
    consumed = arg;
    return valid && best <= budget;
}

// The same verification inside the scan kernel (fz_scan_kernel<..., WF = true>, in-memory searches): queued candidates,
// 64 / GW at a time on GW lanes each, straight from the queue — no hit list, no second kernel and no gap between the two.
// One pass = one memory round trip (the candidates' windows into LDS), the exact n-gram test out of LDS (the scan's filter
// compared a hash), the two expansions, one atomic for the records of the pass.
// (Measured and not kept: right and left expansion of a candidate side by side on two lane groups, the left one with the
// full budget and the pick narrowed afterwards — the rows of a pass drop from max(right) + max(left) over its candidates
// to max(right, left), but a pass takes half the candidates: 0.2128 vs 0.2141 ms where candidates are rare (configs[3a]),
// 0.847 vs 0.605 ms where they are dense and leave after a few rows (1 GiB DNA, m = 40).)
// `have` / `code` are uniform inside a lane group.  Returns the number of confirmed, in-range n-gram hits of the pass.
template <int GW>
__device__ __forceinline__ uint32_t fz_wf_pass(const uint8_t *__restrict__ buf, const FzScanArgs &a, const uint8_t *lds0,
                                               const uint8_t *pat_lds, uint8_t *gwin, bool have, uint32_t code,
                                               FzRec *__restrict__ recs, unsigned long long *__restrict__ counters) {
    static_assert(GW < 64, "one ballot word holds several lane groups");
    const uint32_t lane = fz_lane();
    const uint32_t grp = lane / (uint32_t)GW, gl = lane % (uint32_t)GW;
    uint32_t blk = 0;
    const uint64_t local = fz_code_local(code, blk);
    const uint64_t idx = a.geom.buf_off + local;
    const uint32_t s = (a.g0 + blk) * a.L;
    FzSeg sg;
    sg.sa = 0; sg.se = a.geom.n; sg.j = 0; sg.ok = 1;
    bool valid = have && fz_hit_in_range(a, blk, idx, sg);
    if (!__ballot(valid)) return 0;
    // the candidate's window [wlo, whi), staged as plain bytes: byte g of the sequence at gwin[g - wbase]
    uint64_t whi = 0, wbase = 0;
    if (valid) {
        whi = idx - s + a.m + a.k;
        const uint64_t lim = a.geom.buf_off + a.geom.buf_len;
        if (whi > lim) whi = lim;
        if (whi > sg.se) whi = sg.se;
        wbase = fz_window_base(a, fz_window_lo(a, idx, s, sg.sa));
    }
    const uint32_t nd = valid ? (uint32_t)((whi - wbase + 3) >> 2) : 0u;
    const uint8_t *src = buf + (int64_t)(wbase - a.geom.buf_off);
    for (uint32_t d0 = 0; d0 < a.win_dwords; d0 += 4u * (uint32_t)GW) {       // four loads in flight per lane, then four stores
        uint32_t v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t dd = d0 + j * (uint32_t)GW + gl;
            v[j] = dd < nd ? *reinterpret_cast<const uint32_t *>(src + (size_t)dd * 4) : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t dd = d0 + j * (uint32_t)GW + gl;
            if (dd < nd) reinterpret_cast<uint32_t *>(gwin)[dd] = v[j];
        }
    }
    fz_wave_lds_sync();
    // exact n-gram test: lane gl compares the bytes gl, gl + GW, ...
    bool differs = false;
    if (valid) {
        const uint8_t *tn = gwin + (uint32_t)(idx - wbase), *pn = pat_lds + s;
        for (uint32_t b = gl; b < a.L; b += (uint32_t)GW) differs = differs || tn[b] != pn[b];
    }
    const unsigned long long group_bits = (((1ull << GW) - 1ull) << (grp * (uint32_t)GW));
    if (__ballot(differs) & group_bits) valid = false;
    const unsigned long long vm = __ballot(valid && gl == 0);
    if (!vm) { fz_wave_lds_sync(); return 0; }
    const int prel = (int)(pat_lds - lds0), wrel = (int)(gwin - lds0);
    auto lds_of = [&](uint64_t gidx) -> int { return valid ? wrel + (int)(int64_t)(gidx - wbase) : 0; };
    // right: p[s+L:] vs t[idx+L : min(se, idx-s+m+k)]
    uint64_t rbeg = idx + a.L, rend = idx + a.m + a.k - s;
    if (rend > sg.se) rend = sg.se;
    if (rbeg > sg.se) rbeg = sg.se;
    if (rend < rbeg) rend = rbeg;
    const uint32_t rwin = (uint32_t)(rend - rbeg), rlen = a.m - s - a.L;
    uint32_t dR = 0, r = 0, dL = 0, l = 0;
    const uint32_t cellr = fz_wf_rows<GW>(lds0, gl, a.k, prel + (int)(s + a.L), 1, rlen, lds_of(rbeg), 1, rwin, a.k, valid);
    const bool ok1 = fz_wf_pick<GW>(cellr, gl, a.k, rlen, rwin, a.k, valid, dR, r);
    // left: reversed p[:s] vs reversed t[max(sa, idx-s-(k-dR)) : idx], budget k - dR
    const uint32_t bl = ok1 ? a.k - dR : 0u;
    const uint64_t want = (uint64_t)s + bl;
    const uint64_t lbeg = (idx - sg.sa > want) ? idx - want : sg.sa;
    const uint32_t lwin = ok1 ? (uint32_t)(idx - lbeg) : 0u;
    const uint32_t celll = fz_wf_rows<GW>(lds0, gl, a.k, prel + (int)s - 1, -1, s, lds_of(idx) - 1, -1, lwin, bl, ok1);
    const bool ok = fz_wf_pick<GW>(celll, gl, a.k, s, lwin, bl, ok1, dL, l) && gl == 0;
    const unsigned long long mask = __ballot(ok);
    if (mask) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&counters[1], (unsigned long long)__popcll(mask));
        base = fz_bcast64(base);
        const unsigned long long slot = base + fz_rank(mask);
        if (ok && slot < a.rec_cap) {
            FzRec rec;
            rec.key = fz_hit_pack(a.g0 + blk, idx); rec.l = l; rec.r = r; rec.dist = dL + dR; rec.aux = sg.j;
            recs[slot] = rec;
        }
    }
    fz_wave_lds_sync();
    return (uint32_t)__popcll(vm);
}

// The flush of the fused lane-per-cell form — ONE call site in the scan kernel (a second inlined copy of the pass costs the
// hot loop its registers).  In the middle of the scan (`pooled` = false: this wave's queue filled up) the wave works through
// its own queue.  At the end of the workgroup's life (`pooled` = true, every wave gets here exactly once) what the four waves
// still hold is pooled, as fz_pooled_flush does for the lane-per-candidate form, and dealt out 64 / GW candidates at a time,
// pass p to wave p mod 4: the n-gram hits of one near match sit a few bytes apart, i.e. in ONE wave's queue, and would take
// that wave several passes while the other three have none.
// `area` = first wave's area (queue first), `per_wave` = bytes per wave area, `fills` = four LDS dwords.
template <int GW>
__device__ __forceinline__ uint32_t fz_flush_wf(const uint8_t *__restrict__ buf, const FzScanArgs &a, const uint8_t *lds0,
                                                const uint8_t *pat_lds, const FzWaveLds &w, const uint8_t *area, uint32_t per_wave,
                                                volatile uint32_t *fills, uint32_t wave, uint32_t qn, bool pooled,
                                                FzRec *__restrict__ recs, unsigned long long *__restrict__ counters) {
    constexpr uint32_t NH = 64u / (uint32_t)GW;                         // candidates per pass
    const uint32_t lane = fz_lane();
    const uint32_t grp = lane / (uint32_t)GW;
    uint32_t n0 = qn, n1 = 0, n2 = 0, n3 = 0, first = 0, step = NH;
    const uint8_t *q0 = reinterpret_cast<const uint8_t *>(w.queue);
    if (pooled) {
        if (lane == 0) fills[wave] = qn;
        __syncthreads();
        n0 = fills[0]; n1 = fills[1]; n2 = fills[2]; n3 = fills[3];
        first = wave * NH;
        step = FZ_WAVES_PER_BLOCK * NH;
        q0 = area;
    } else {
        fz_wave_lds_sync();
    }
    const uint32_t total = n0 + n1 + n2 + n3;
    uint8_t *gwin = reinterpret_cast<uint8_t *>(w.win) + grp * (a.win_dwords * 4u + 16u);
    uint32_t confirmed = 0;
    // (Measured: a pass with the hits of one near match — different blocks, so max(right) + max(left) ~ 100 rows of a
    // 64-byte pattern — takes 17 - 23 us, ~400 cycles per row, next to six streaming waves per SIMD; s_setprio(3)
    // around the passes changed nothing: the rows wait for their own dependent instructions, not for issue slots.)
    for (uint32_t e0 = first; e0 < total; e0 += step) {
        uint32_t li = e0 + grp, ow = 0;
        const bool have = li < total;
        if (li >= n0) { li -= n0; ow = 1; if (li >= n1) { li -= n1; ow = 2; if (li >= n2) { li -= n2; ow = 3; } } }
        const uint32_t code = have ? reinterpret_cast<const uint32_t *>(q0 + ow * per_wave)[li] : 0u;
        confirmed += fz_wf_pass<GW>(buf, a, lds0, pat_lds, gwin, have, code, recs, counters);
    }
    return confirmed;
}

// Levenshtein verification of a hit list, GW lanes per hit (64 / GW hits per wave at a time): right
// expansion, then left with what the right one left of the budget (levenshtein_ngram.py:177-189).
// (Measured and dropped: right and left expansions of a hit side by side on two lane groups — half the
// rows per wave but twice the workgroups, 0.031 vs 0.029 ms: the rows are ~11 us of this kernel, the rest
// is launch, two dependent load round trips and the finish tickets.)
// Record slot = hit number (x candidate segment): no slot atomics — all waves of this kernel finish at
// about the same time, and 1400 atomics on one counter word took longer than the DP rows.  Slots of hits
// that did not verify carry FZ_REC_NONE; the host drops them.  Long hit lists (more than FZ_WF_COMPACT_MIN
// slots) append their records instead, one atomic per pass that verified something: a slot per hit of a
// dense list is megabytes of empty records for the host to copy and skip (2.4e6 hits of 1 GiB of DNA at
// m = 54, k = 8: 57 MB per search, 16 of the call's 19 ms).
// Dynamic LDS: pattern + per-wave window areas (one contiguous byte window per hit).
template <int GW>
__global__ __launch_bounds__(1024) void fz_verify_wf_kernel(const uint8_t *__restrict__ buf, const FzScanArgs a,
                                                            const uint64_t *__restrict__ hits, FzRec *__restrict__ recs,
                                                            unsigned long long *__restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t NH = 64u / (uint32_t)GW;                         // hits per wave
    const uint32_t mpad = (a.m + 15u) & ~15u;
    uint8_t *pat_lds = smem + 16;                                       // 16 bytes of slack below p[0] (reversed reads)
    fz_copy_pattern(pat_lds, a, threadIdx.x, blockDim.x);
    __syncthreads();
    const uint32_t lane = fz_lane();
    const uint32_t grp = lane / (uint32_t)GW, gl = lane % (uint32_t)GW;
    const uint32_t wbytes = a.win_dwords * 4u;
    uint8_t *gwin = smem + 16 + mpad + 16 + ((threadIdx.x >> 6) * NH + grp) * (wbytes + 16u);
    unsigned long long nh = counters[0];
    if (nh > a.hit_cap) nh = a.hit_cap;
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t ncand = fz_segment_candidates(a.geom);
    // workgroups without a hit leave at once and take no finish ticket (a ticket is an atomic on one word)
    const uint64_t per_wg = (uint64_t)NH * (blockDim.x >> 6);
    uint32_t active_wgs = (uint32_t)((nh + per_wg - 1) / per_wg < gridDim.x ? (nh + per_wg - 1) / per_wg : gridDim.x);
    if (active_wgs == 0) active_wgs = 1;
    if (blockIdx.x >= active_wgs) return;
    const bool compact = nh * ncand > FZ_WF_COMPACT_MIN;               // (uniform: every wave takes the same form)
    for (uint64_t q0 = wave * NH; q0 < nh; q0 += waves * NH) {
        const uint64_t q = q0 + grp;
        const bool have = q < nh;
        const uint64_t hit = have ? hits[q] : 0;
        const uint32_t g = fz_hit_block(hit);
        const uint64_t idx = fz_hit_index(hit);
        const uint32_t s = g * a.L;
        for (uint32_t c = 0; c < ncand; ++c) {
            const FzSeg sg = fz_segment(a.geom, idx, c);
            const bool valid = have && fz_hit_in_range_s(a, s, idx, sg);
            const unsigned long long slot = q * ncand + c;
            if (!compact && have && !valid && gl == 0 && slot < a.rec_cap) recs[slot].dist = FZ_REC_NONE;   // not a hit of this segment
            if (!__ballot(valid)) continue;
            // the hit's window [wlo, whi), staged as plain bytes: byte g of the sequence at gwin[g - wbase]
            uint64_t wlo = 0, whi = 0, wbase = 0;
            if (valid) {
                const uint64_t reach = (uint64_t)s + a.k;
                wlo = idx - sg.sa > reach ? idx - reach : sg.sa;
                if (wlo < a.geom.buf_off) wlo = a.geom.buf_off;
                whi = idx - s + a.m + a.k;
                const uint64_t lim = a.geom.buf_off + a.geom.buf_len;
                if (whi > lim) whi = lim;
                if (whi > sg.se) whi = sg.se;
                wbase = a.geom.buf_off + ((wlo - a.geom.buf_off) & ~(uint64_t)3);
            }
            const uint32_t nd = valid ? (uint32_t)((whi - wbase + 3) >> 2) : 0u;
            const int64_t lbase = (int64_t)(wbase - a.geom.buf_off);
            for (uint32_t dd = gl; dd < nd; dd += (uint32_t)GW)
                reinterpret_cast<uint32_t *>(gwin)[dd] = *reinterpret_cast<const uint32_t *>(buf + lbase + (int64_t)dd * 4);
            fz_wave_lds_sync();
            // LDS byte offsets relative to smem (invalid groups read offset 0)
            const int wrel = (int)(gwin - smem);
            auto lds_of = [&](uint64_t gidx) -> int { return valid ? wrel + (int)(int64_t)(gidx - wbase) : 0; };
            // right: p[s+L:] vs t[idx+L : min(se, idx-s+m+k)]
            uint64_t rbeg = idx + a.L, rend = idx + a.m + a.k - s;
            if (rend > sg.se) rend = sg.se;
            if (rbeg > sg.se) rbeg = sg.se;
            if (rend < rbeg) rend = rbeg;
            const uint32_t rwin = (uint32_t)(rend - rbeg), rlen = a.m - s - a.L;
            uint32_t dR = 0, r = 0, dL = 0, l = 0;
            const uint32_t cellr = fz_wf_rows<GW>(smem, gl, a.k, (int)(pat_lds - smem) + (int)(s + a.L), 1, rlen, lds_of(rbeg), 1, rwin,
                                                  a.k, valid);
            const bool ok1 = fz_wf_pick<GW>(cellr, gl, a.k, rlen, rwin, a.k, valid, dR, r);
            // left: reversed p[:s] vs reversed t[max(sa, idx-s-(k-dR)) : idx], budget k - dR
            const uint32_t bl = ok1 ? a.k - dR : 0u;
            const uint64_t want = (uint64_t)s + bl;
            const uint64_t lbeg = (idx - sg.sa > want) ? idx - want : sg.sa;
            const uint32_t lwin = ok1 ? (uint32_t)(idx - lbeg) : 0u;
            const uint32_t celll = fz_wf_rows<GW>(smem, gl, a.k, (int)(pat_lds - smem) + (int)s - 1, -1, s, lds_of(idx) - 1, -1, lwin,
                                                  bl, ok1);
            const bool ok = fz_wf_pick<GW>(celll, gl, a.k, s, lwin, bl, ok1, dL, l);
            if (!compact) {
                if (valid && gl == 0 && slot < a.rec_cap) {
                    FzRec rec;
                    rec.key = hit; rec.l = l; rec.r = r; rec.dist = ok ? dL + dR : FZ_REC_NONE; rec.aux = sg.j;
                    recs[slot] = rec;
                }
            } else {
                const bool mine = ok && gl == 0;
                const unsigned long long mask = __ballot(mine);
                if (mask) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(&counters[1], (unsigned long long)__popcll(mask));
                    base = fz_bcast64(base);
                    const unsigned long long at = base + fz_rank(mask);
                    if (mine && at < a.rec_cap) {
                        FzRec rec;
                        rec.key = hit; rec.l = l; rec.r = r; rec.dist = dL + dR; rec.aux = sg.j;
                        recs[at] = rec;
                    }
                }
            }
            fz_wave_lds_sync();
        }
    }
    // the record count the host sees = number of slots (or of appended records); only workgroups that had hits take a finish ticket
    if (!compact && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counters[1], nh * ncand);
    fz_finish_launch(a, counters, reinterpret_cast<uint32_t *>(smem), active_wgs);
}

// ---------------------------------------------------------------------------------------------
// fz_verify_big_kernel<CPL> — verification without size limits: ONE WAVE PER HIT, lane-per-DP-cell with CPL band
// cells per lane (2k + 1 <= 64 * CPL cells: budgets up to 1023), pattern and sequence read from HBM / L2 as they
// are (no LDS staging: subsequences up to FZ_MAX_M_ANY).  Used where the other verifications do not fit: patterns
// longer than the kernel-argument block, budgets above FZ_MAX_K, windows or score rings beyond LDS
// (levenshtein_ngram.py:159-198 has no limit on either).  Same table as fz_expand / c_expand_*
// (_levenshtein_ngrams.pyx:9-154): lane gl owns the cells x = gl * CPL + c of a row, cell x of row i is
// D[i][i + x - K]; the upper neighbour is cell x + 1 of the previous row (the next lane's first cell: one DPP wave
// shift), the left-neighbour recurrence v[x] = min_{e <= x} (a[e] + x - e) is a prefix-min of a[e] - e: sequential
// inside the lane, one exclusive wave scan (DPP row shifts + row broadcasts) across lanes.  Cells beyond 2K are
// computed as well (a wider band is still exact).  The 64 pattern characters and the 64 characters entering the
// band's right edge of the next 64 rows are fetched with one coalesced load each and handed out by v_readlane.
__device__ __forceinline__ uint32_t fz_dpp_wave_shl1(uint32_t v, uint32_t edge) {     // lane l <- lane l + 1; lane 63 <- edge
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t fz_dpp_wave_shr1(uint32_t v, uint32_t edge) {     // lane l <- lane l - 1; lane 0 <- edge
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xf, 0xf, false);
}
// inclusive prefix-min over the 64 lanes (lanes without a source keep their value)
__device__ __forceinline__ uint32_t fz_wave_incl_min(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return v;
}

// One bounded expansion by the whole wave; every argument is wave-uniform.  sub(i) = psub[i * sstep] (i < sublen),
// win(j) = pwin[j * wstep] (j < winlen), both in global memory.  -> (ok, dist, consumed) as fz_expand.
template <int CPL>
__device__ __forceinline__ bool fz_big_expand(const uint8_t *psub, int sstep, uint32_t sublen, const uint8_t *pwin, int wstep,
                                              uint32_t winlen, uint32_t K, uint32_t budget, uint32_t &dist, uint32_t &consumed) {
    if (sublen == 0) { dist = 0; consumed = 0; return true; }       // pyx:28-30
    constexpr uint32_t INF = 0x3fffffffu;
    constexpr uint32_t NX = 64u * (uint32_t)CPL;
    const uint32_t lane = fz_lane();
    const uint32_t x0 = lane * (uint32_t)CPL;
    uint32_t cell[CPL], chr[CPL];
    int jb = (int)x0 - (int)K;                                       // column of cell 0 of this lane in row i: jb + c (row 0 now)
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int j = jb + c;
        cell[c] = (j >= 0 && (uint32_t)j <= winlen) ? (uint32_t)j : INF;                       // row 0: D[0][j] = j
        chr[c] = (j >= 0 && (uint32_t)j < winlen) ? (uint32_t)pwin[(int64_t)j * wstep] : 0x100u;   // row 1 compares win[j]
    }
    uint32_t pvec = 0, wvec = 0;
    for (uint32_t i = 1; i <= sublen; ++i) {
        const uint32_t r = (i - 1u) & 63u;
        if (r == 0) {
            const uint32_t pi = i - 1u + lane;                                                // pattern characters of rows i .. i + 63
            pvec = pi < sublen ? (uint32_t)psub[(int64_t)pi * sstep] : 0u;
            const int64_t wj = (int64_t)(i + lane) + (int64_t)NX - 1 - (int64_t)K;           // right-edge character after row i + lane
            wvec = (wj >= 0 && wj < (int64_t)winlen) ? (uint32_t)pwin[wj * wstep] : 0x100u;
        }
        const uint32_t pc = (uint32_t)__builtin_amdgcn_readlane((int)pvec, (int)r);
        const uint32_t edge = (uint32_t)__builtin_amdgcn_readlane((int)wvec, (int)r);
        jb += 1;
        const uint32_t up_last = fz_dpp_wave_shl1(cell[0], INF);                             // D[i-1][..] of the next lane's first cell
        // a[c] = min(diag + cost, up + 1), forced at column 0 and outside the table; t[c] = running min of a[e] + (NX - e)
        uint32_t run = 0xffffffffu;
        uint32_t t[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = jb + c;
            const uint32_t up = (c + 1 < CPL) ? cell[(c + 1 < CPL) ? c + 1 : 0] : up_last;
            uint32_t v = min(cell[c] + (chr[c] != pc ? 1u : 0u), up + 1u);
            if (j == 0) v = i;                                                               // D[i][0] = i
            if (j < 0 || (uint32_t)j > winlen) v = INF;
            run = min(run, v + (NX - (x0 + (uint32_t)c)));
            t[c] = run;
        }
        const uint32_t incl = fz_wave_incl_min(run);
        const uint32_t excl = fz_dpp_wave_shr1(incl, 0xffffffffu);
        uint32_t any_low = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = jb + c;
            uint32_t v = min(excl, t[c]) - (NX - (x0 + (uint32_t)c));
            v = min(v, INF);
            if (j < 0 || (uint32_t)j > winlen) v = INF;
            cell[c] = v;
            any_low |= (v <= budget) ? 1u : 0u;
        }
        // next row's characters: every cell takes its right neighbour's
        const uint32_t in = fz_dpp_wave_shl1(chr[0], edge);
#pragma unroll
        for (int c = 0; c + 1 < CPL; ++c) chr[c] = chr[c + 1];
        chr[CPL - 1] = in;
        // row minima never decrease: once no cell is within the budget nothing can pass (pyx:61-65)
        if ((i & 3u) == 0u && !__ballot(any_low != 0u)) return false;
    }
    // bottom row: min over the columns 1 .. winlen, the LARGEST column among equals, from the column-0 baseline
    uint32_t key = 0xffffffffu;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int j = jb + c;
        if (j >= 1 && (uint32_t)j <= winlen) key = min(key, (min(cell[c], 0xfffffu) << 12) | (NX - 1u - (x0 + (uint32_t)c)));
    }
    key = fz_wave_incl_min(key);
    key = (uint32_t)__builtin_amdgcn_readlane((int)key, 63);
    uint32_t best = sublen, arg = 0;
    if (key != 0xffffffffu && (key >> 12) <= sublen) {
        best = key >> 12;
        arg = (uint32_t)((int)sublen + (int)(NX - 1u - (key & 0xfffu)) - (int)K);
    }
    dist = best;
    consumed = arg;
    return best <= budget;
}

template <int CPL>
__global__ __launch_bounds__(64) void fz_verify_big_kernel(const uint8_t *__restrict__ buf, const FzScanArgs a,
                                                           const uint64_t *__restrict__ hits, FzRec *__restrict__ recs,
                                                           unsigned long long *__restrict__ counters) {
    __shared__ uint32_t flag;
    const uint32_t lane = fz_lane();
    const uint8_t *pat = reinterpret_cast<const uint8_t *>(a.pat_g);   // the host stages the pattern in HBM for this kernel
    unsigned long long nh = counters[0];
    if (nh > a.hit_cap) nh = a.hit_cap;
    const uint32_t ncand = fz_segment_candidates(a.geom);
    for (uint64_t q = blockIdx.x; q < nh; q += gridDim.x) {
        const uint64_t hit = hits[q];
        const uint32_t g = fz_hit_block(hit);
        const uint64_t idx = fz_hit_index(hit);
        const uint32_t s = g * a.L;
        for (uint32_t c = 0; c < ncand; ++c) {
            const FzSeg sg = fz_segment(a.geom, idx, c);
            if (!fz_hit_in_range_s(a, s, idx, sg)) continue;                               // wave-uniform
            FzRec rec;
            bool ok;
            if (a.mode == FZ_MODE_SUBS) {
                // Hamming distance of the window [idx - s, idx - s + m) (_substitutions_only_ngrams_template.h:103-121)
                const uint8_t *t = buf + (int64_t)(idx - s - a.geom.buf_off);
                uint32_t nd = 0;
                for (uint32_t q0 = 0; q0 < a.m; q0 += 4096u) {
                    for (uint32_t qq = q0 + lane; qq < a.m && qq < q0 + 4096u; qq += 64u) nd += (pat[qq] != t[qq]) ? 1u : 0u;
                    if (!__ballot(nd <= a.k)) break;                                        // some lane alone is over the budget
                }
                uint32_t tot = nd;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) tot += (uint32_t)__shfl_xor((int)tot, d, 64);
                ok = tot <= a.k;                                                            // (after an early exit tot is partial, but > k)
                rec.l = s; rec.r = a.m - s - a.L; rec.dist = tot;
            } else {
                // levenshtein_ngram.py:177-198: right expansion with budget k, then left with what is left of it
                const uint32_t rlen = a.m - s - a.L;
                uint64_t rbeg = idx + a.L, rend = idx + a.m + a.k - s;
                if (rend > sg.se) rend = sg.se;
                if (rbeg > sg.se) rbeg = sg.se;
                if (rend < rbeg) rend = rbeg;
                uint32_t dR = 0, r = 0, dL = 0, l = 0;
                ok = fz_big_expand<CPL>(pat + s + a.L, 1, rlen, buf + (int64_t)(rbeg - a.geom.buf_off), 1, (uint32_t)(rend - rbeg),
                                        a.k, a.k, dR, r);
                if (ok) {
                    const uint32_t bl = a.k - dR;
                    const uint64_t want = (uint64_t)s + bl;
                    const uint64_t lbeg = (idx - sg.sa > want) ? idx - want : sg.sa;
                    ok = fz_big_expand<CPL>(pat + s - 1, -1, s, buf + (int64_t)(idx - a.geom.buf_off) - 1, -1, (uint32_t)(idx - lbeg),
                                            a.k, bl, dL, l);
                }
                rec.l = l; rec.r = r; rec.dist = dL + dR;
            }
            if (ok && lane == 0) {
                const unsigned long long slot = atomicAdd(&counters[1], 1ull);
                rec.key = hit;
                rec.aux = sg.j;
                if (slot < a.rec_cap) recs[slot] = rec;
            }
        }
    }
    fz_finish_launch(a, counters, &flag);
}

// ---------------------------------------------------------------------------------------------
// K4  fz_generic_kernel: the generic (mixed-limit) search's per-hit verification — the greedy
// candidate-set automaton of generic_search.py:57-177 run on the window
// seq[max(0, idx-s-k) : idx-s+m+k] of every exact n-gram hit (generic_search.py:222-237).
// One WAVE per hit: the candidate list lives in LDS and its 64-wide slices are advanced by the 64
// lanes; successor candidates and matches are written through wave prefix sums, so both lists keep
// exactly the reference's order (its emitted *list*, not just the set, is reproduced).  The
// character loop is inherently sequential; parallelism comes from hits x candidates.
#ifndef FZ_LP_PAIR
#define FZ_LP_PAIR 1                                       // the per-hit automaton steps two 64-candidate slices per trip
#endif
#ifndef FZ_GEN_MCAP
#define FZ_GEN_MCAP 128                                    // match-buffer entries per wave (1 KB: with 256-entry candidate lists
                                                           // 24 waves per CU are resident, every hit of configs[3b] at once)
#endif

// Inclusive prefix sum over the 64 lanes on DPP: four row shifts scan the 16-lane rows, row_bcast:15 / :31 carry
// the row totals across (6 VALU adds; the ds_bpermute form — six __shfl_up — paid six LDS round trips per call,
// and the automaton calls it once per 64 candidates per window character).
__device__ __forceinline__ uint32_t fz_wave_incl_scan(uint32_t v) {
    // update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes without a source keep `old` (0 here)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2 and 3
    return v;
}

// KIND (FzLpKind) and HBM_LISTS are compile-time: the per-hit instance carries none of the tiled Levenshtein
// automaton's code, and with the lists in LDS their accesses are ds_ instructions instead of flat ones.
template <int KIND, bool HBM_LISTS>
__global__ __launch_bounds__(64) void fz_lp_kernel(const uint8_t *__restrict__ buf, const FzScanArgs a,
                                                   const uint64_t *__restrict__ hits, uint64_t n_items,
                                                   FzGenRec *__restrict__ recs,
                                                   unsigned long long *__restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x;
    const uint32_t mpad = (a.m + 15u) & ~15u;
    const uint32_t wmax = a.m + 2u * a.k + a.lp_starts;
    const uint32_t wpad = (wmax + 15u) & ~15u;
    uint8_t *pat = smem;
    uint8_t *win = smem + mpad;
    // Candidate lists: LDS normally; for inputs whose candidate sets outgrow it (two-letter alphabets with
    // large budgets) per-workgroup lists in HBM.  One wave per workgroup, so the workgroup-scope fences
    // of fz_wave_lds_sync order its global accesses just as they order the LDS ones.
    FzGCand *cur = HBM_LISTS ? reinterpret_cast<FzGCand *>(a.cand_scratch) + (size_t)blockIdx.x * 2u * a.cand_cap
                             : reinterpret_cast<FzGCand *>(smem + mpad + wpad);
    FzGCand *nxt = cur + a.cand_cap;
    uint64_t *mbuf = reinterpret_cast<uint64_t *>(smem + mpad + wpad + (HBM_LISTS ? 0u : 2u * a.cand_cap * (uint32_t)sizeof(FzGCand)));
    fz_copy_pattern(pat, a, lane, 64u);
    fz_wave_lds_sync();
    auto patf = [&](uint32_t i) -> uint8_t { return pat[i]; };
    constexpr bool per_hit = KIND == FZ_LP_GENERIC_HIT;
    constexpr bool lev = KIND == FZ_LP_LEV_SEQ;

    unsigned long long nitems = n_items;
    if (per_hit) { nitems = counters[0]; if (nitems > a.hit_cap) nitems = a.hit_cap; }
    const uint32_t ncand = per_hit ? fz_segment_candidates(a.geom) : 1u;
    // device-side ordering (fz_gen_order_kernel / fz_gen_scatter_kernel below): every hit leaves its row count
    const bool order = per_hit && a.gen_order != 0;
    unsigned long long *order_first = reinterpret_cast<unsigned long long *>(a.gen_order);
    uint32_t *order_count = reinterpret_cast<uint32_t *>(order_first + FZ_GEN_ORDER_MAX);
    // window table (FzGenDedup, fz_device.h): hits that share a window run the automaton once
    const bool dedup = per_hit && a.gen_dedup != 0 && nitems <= FZ_GEN_ORDER_MAX && ncand == 1u;
    const FzGenDedup dd(a.gen_dedup);
    for (uint64_t qc = blockIdx.x; qc < nitems * ncand; qc += gridDim.x) {
        if ((a.flags & FZ_FLAG_ANY) && __hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;   // has_near_match_*
        const uint64_t q = qc / ncand;
        uint64_t key_base, w0, w1;
        uint32_t spawn_len, seg_j = 0;
        bool flush_end;
        if (per_hit) {
            const uint64_t hit = hits[q];
            const uint32_t s = fz_hit_block(hit) * a.L;
            const uint64_t idx = fz_hit_index(hit);
            const FzSeg sg = fz_segment(a.geom, idx, (uint32_t)(qc % ncand));
            if (!fz_hit_in_range_s(a, s, idx, sg)) {                   // wave-uniform: one hit per wave
                if (order && q < FZ_GEN_ORDER_MAX && lane == 0) { order_first[q] = 0; order_count[q] = 0; }
                continue;
            }
            if (dedup) {
                // the scan entered the hit into the window table: only the window's leader (smallest block) runs
                const uint32_t sl = dd.wslot[q];
                if (sl != FZ_GEN_DEDUP_NONE && dd.leader(sl) != (uint32_t)q) {
                    if (order && lane == 0) { order_first[q] = 0; order_count[q] = 0; }
                    continue;
                }
            }
            const uint64_t reach = (uint64_t)s + a.k;
            w0 = idx - sg.sa > reach ? idx - reach : sg.sa;            // generic_search.py:231
            w1 = idx - s + a.m + a.k;
            if (w1 > sg.se) w1 = sg.se;
            spawn_len = (uint32_t)(w1 - w0);
            flush_end = true;                                          // the window IS the sequence there
            key_base = hit;
            seg_j = sg.j;
        } else {
            w0 = a.geom.own_lo + q * a.lp_starts;
            const uint64_t own_end = a.geom.own_hi < a.geom.n ? a.geom.own_hi : a.geom.n;
            spawn_len = (uint32_t)((own_end - w0) < a.lp_starts ? (own_end - w0) : a.lp_starts);
            w1 = w0 + spawn_len + a.m + a.k;
            if (w1 > a.geom.n) w1 = a.geom.n;
            flush_end = w1 == a.geom.n;     // candidates alive at the true sequence end are flushed;
            key_base = w0;                  // elsewhere every candidate of this tile has died by w1
        }
        const uint32_t wlen = (uint32_t)(w1 - w0);
        for (uint32_t i = lane; i < wlen; i += 64u) win[i] = buf[(w0 - a.geom.buf_off) + i];
        fz_wave_lds_sync();

        uint32_t ncur = 0, mb = 0, mseq = 0;
        bool overflow = false;
        // FZ_FLAG_FOLD (fz_generic_ngrams_consolidated): what leaves the kernel is not the hit's matches but (hull, best
        // match) pairs — every match that overlaps the running hull of the hit's earlier matches is folded into it
        // (the group test of common.py:150-159; "best" = smallest distance, then longest, then smallest start: a total
        // order, so the fold order does not matter).  The matches of one hit nearly always overlap: ~1 pair per hit
        // instead of ~34 rows (configs[3b]: 6e3 pairs instead of 2.1e5 rows cross PCIe), and the host only merges hulls.
        const bool fold = per_hit && (a.flags & FZ_FLAG_FOLD);
        bool f_have = false;
        uint32_t f_lo = 0, f_hi = 0, f_k1 = 0, f_k2 = 0, f_pairs = 0;   // hull [lo, hi), best = (dist << 16 | 0xffff - len, start)
        auto emit_pair = [&]() {
            if (lane == 0) {
                // (measured: without this atomic — slot = hit number — the kernel takes the same 0.164 ms: the slot counter is
                //  not what the automaton waits for)
                // Window table: this hit stands for every hit of its window.  Their equal matches fall into one overlap group
                // (the smallest block — this hit's — is the one a consolidation keeps) EXCEPT zero-length matches, which
                // overlap nothing, not even their own copies (common.py:152-153): a pair with an empty hull is emitted once
                // per member of the window, under that member's key.
                const uint32_t sl = dedup ? dd.wslot[q] : FZ_GEN_DEDUP_NONE;
                uint32_t copies = 1u;
                if (sl != FZ_GEN_DEDUP_NONE && f_lo == f_hi) { copies = dd.nmem[sl]; copies = copies < FZ_GEN_DEDUP_MEMBERS ? copies : FZ_GEN_DEDUP_MEMBERS; }
                for (uint32_t cpy = 0; cpy < copies; ++cpy) {
                    const unsigned long long slot = atomicAdd(&counters[1], 1ull);
                    if (slot < a.rec_cap) {
                        const uint32_t len = 0xffffu - (f_k1 & 0xffffu);
                        FzGenRec r;
                        r.key = copies > 1u ? hits[dd.mem[sl * FZ_GEN_DEDUP_MEMBERS + cpy]] : key_base;
                        r.seq = f_pairs; r.se = f_k2 | ((f_k2 + len) << 16); r.dist = f_k1 >> 16; r.win = f_lo | (f_hi << 16);
                        recs[slot] = r;
                    }
                }
            }
            ++f_pairs;
            f_have = false;
        };
        auto wave_min = [&](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)fz_wave_incl_min(v), 63); };
        auto fold_matches = [&]() {
            fz_wave_lds_sync();
            for (uint32_t e0 = 0; e0 < mb; e0 += 64u) {
                const bool have_row = e0 + lane < mb;
                const uint64_t v = have_row ? mbuf[e0 + lane] : 0ull;
                const uint32_t rs = (uint32_t)v & 0xffffu, re = ((uint32_t)v >> 16) & 0xffffu, rd = (uint32_t)(v >> 32) & 0xffffu;
                const uint32_t k1 = (rd << 16) | (0xffffu - (re - rs));
                unsigned long long pending = __ballot(have_row);
                while (pending) {
                    if (!f_have) {                                 // a new hull starts with the first row that is left
                        const uint32_t l0 = (uint32_t)__builtin_ctzll(pending);
                        f_lo = (uint32_t)__builtin_amdgcn_readlane((int)rs, (int)l0);
                        f_hi = (uint32_t)__builtin_amdgcn_readlane((int)re, (int)l0);
                        f_k1 = (uint32_t)__builtin_amdgcn_readlane((int)k1, (int)l0);
                        f_k2 = f_lo;
                        f_have = true;
                        pending &= pending - 1ull;
                        continue;
                    }
                    const bool mine = ((pending >> lane) & 1ull) && !(re <= f_lo || rs >= f_hi);
                    const unsigned long long ov = __ballot(mine);
                    if (!ov) { emit_pair(); continue; }            // nothing left overlaps this hull: it is complete for now
                    const uint32_t mlo = wave_min(mine ? rs : 0xffffffffu), mhi = ~wave_min(mine ? ~re : 0xffffffffu);
                    const uint32_t mk1 = wave_min(mine ? k1 : 0xffffffffu);
                    const uint32_t mk2 = wave_min((mine && k1 == mk1) ? rs : 0xffffffffu);
                    f_lo = mlo < f_lo ? mlo : f_lo;
                    f_hi = mhi > f_hi ? mhi : f_hi;
                    if (mk1 < f_k1 || (mk1 == f_k1 && mk2 < f_k2)) { f_k1 = mk1; f_k2 = mk2; }
                    pending &= ~ov;
                }
            }
            mseq += mb;
            mb = 0;
            fz_wave_lds_sync();
        };
        // match buffer entry: se (32) | dist (16) | step-in-window (16)
        auto flush_matches = [&]() {
            if (mb == 0) return;
            if (fold) { fold_matches(); return; }
            fz_wave_lds_sync();
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&counters[1], (unsigned long long)mb);
            base = fz_bcast64(base);
            for (uint32_t e = lane; e < mb; e += 64u) {
                if (base + e < a.rec_cap) {
                    const uint64_t v = mbuf[e];
                    FzGenRec r;
                    r.key = per_hit ? key_base : key_base + (v >> 48);
                    r.seq = mseq + e; r.se = (uint32_t)v; r.dist = (uint32_t)(v >> 32) & 0xffffu; r.win = per_hit ? (order ? (uint32_t)q : seg_j) : (uint32_t)q;
                    recs[base + e] = r;
                }
            }
            mseq += mb;
            mb = 0;
            fz_wave_lds_sync();
        };
        for (uint32_t index = 0; index <= wlen && !overflow; ++index) {
            const bool last = index == wlen;                           // end-of-window flush pass
            if (last && !flush_end) break;
            uint8_t ch = 0;
            uint32_t nnext = 0;
            uint32_t fresh_at = 0xffffffffu;                           // list position of this character's fresh candidate
            if (!last) {
                ch = win[index];
                if (index < spawn_len) {
                    if (!lev) {                                        // generic: fresh candidate appended (py:80)
                      // (per-hit windows: only starts that can still reach the pattern's end, fz_gen_start_useful)
                      if (!per_hit || fz_gen_start_useful(index, wlen, a.m, a.max_dels, a.k)) {
                        if (ncur >= a.cand_cap) { overflow = true; break; }
                        // (start = index, everything else 0) never goes through the list: the lane that owns
                        // position ncur takes it from registers — no LDS store + wait per character
                        fresh_at = ncur;
                        ++ncur;
                      }
                    } else {                                           // Levenshtein: levenshtein.py:75-80
                        uint32_t f = 0xffffffffu;
                        const uint32_t lim = a.k + 1 < a.m ? a.k + 1 : a.m;
                        for (uint32_t i = 0; i < lim; ++i) if (pat[i] == ch) { f = i; break; }
                        if (f != 0xffffffffu) {
                            if (f + 1 == a.m) {                        // immediate match, emitted first
                                if (mb + 1 > FZ_GEN_MCAP) flush_matches();
                                if (lane == 0) mbuf[mb] = (uint64_t)(index | ((index + 1) << 16)) | ((uint64_t)f << 32) | ((uint64_t)index << 48);
                                ++mb;
                            } else {                                   // new candidate goes FIRST
                                if (lane == 0) { FzGCand c; c.start = (uint16_t)index; c.j = (uint16_t)(f + 1); c.l = (uint8_t)f; c.ns = c.ni = c.nd = 0; nxt[0] = c; }
                                nnext = 1;
                            }
                        }
                    }
                }
            }
            const bool more_seq = w0 + index + 1 < a.geom.n;   // tiled Levenshtein mode only
            // the outputs of one slice of 64 candidates go to the lists through wave prefix sums, in list order
            auto emit = [&](const FzGStep &st) -> bool {
                const uint32_t packed = (st.fa + st.fb + st.fc) | ((st.f1 + st.f2) << 16);
                const uint32_t incl = fz_wave_incl_scan(packed);
                const uint32_t tot = __builtin_amdgcn_readlane(incl, 63);
                const uint32_t excl = incl - packed;
                const uint32_t tot_s = tot & 0xffffu, tot_m = tot >> 16;
                if (nnext + tot_s > a.cand_cap) return false;
                if (mb + tot_m > FZ_GEN_MCAP) flush_matches();
                uint2 *nx = reinterpret_cast<uint2 *>(nxt) + nnext + (excl & 0xffffu);
                uint64_t *mp = mbuf + mb + (excl >> 16);
                const uint64_t stamp = (uint64_t)index << 48;
                {
                    if (st.fa) nx[0] = make_uint2(st.a0, st.a1);
                    if (st.fb) nx[st.fa] = make_uint2(st.b0, st.b1);
                    if (st.fc) nx[st.fa + st.fb] = make_uint2(st.c0, st.c1);
                    if (st.f1) mp[0] = (uint64_t)st.m1 | ((uint64_t)st.d1 << 32) | stamp;
                    if (st.f2) mp[st.f1] = (uint64_t)st.m2 | ((uint64_t)st.d2 << 32) | stamp;
                }
                nnext += tot_s;
                mb += tot_m;
                return true;
            };
            if (!last && !lev) {
                // the hot loop (its own copy of `emit`: no register shuffling where the two forms would join).  Every
                // lane steps — positions below cand_cap are readable, lanes past the list step an all-zero candidate
                // and have their five output flags cleared: no divergence
                auto load_step = [&](uint32_t c0, FzGStep &st) {
                    const bool valid = c0 + lane < ncur;
                    uint2 cw = reinterpret_cast<const uint2 *>(cur)[c0 + lane];
                    const bool fresh = c0 + lane == fresh_at;
                    cw.x = valid ? (fresh ? index : cw.x) : 0u;
                    cw.y = valid && !fresh ? cw.y : 0u;
                    fz_generic_step_packed(cw.x, cw.y, ch, index, a.m, patf, a.max_subs, a.max_ins, a.max_dels, a.k, st);
                    const uint32_t vm = valid ? 1u : 0u;
                    st.fa &= vm; st.fb &= vm; st.fc &= vm; st.f1 &= vm; st.f2 &= vm;
                };
                uint32_t c0 = 0;
#if FZ_LP_PAIR
                // Two slices of 64 candidates per trip while the list has more than one: their loads, steps and prefix
                // scans are independent chains the hardware can overlap (a hit's time is this dependent chain, character
                // after character, and the kernel takes as long as its slowest hits — the ones with long lists); only
                // the output offsets of the second slice wait for the first one's totals.  (Lists in LDS only: the
                // second slice may read up to 127 entries past the list, which there is still this wave's LDS.)
                if constexpr (!HBM_LISTS) {
                    for (; c0 + 64u < ncur && !overflow; c0 += 128u) {
                        FzGStep sa, sb;
                        load_step(c0, sa);
                        load_step(c0 + 64u, sb);
                        const uint32_t pa = (sa.fa + sa.fb + sa.fc) | ((sa.f1 + sa.f2) << 16);
                        const uint32_t pb = (sb.fa + sb.fb + sb.fc) | ((sb.f1 + sb.f2) << 16);
                        const uint32_t ia = fz_wave_incl_scan(pa), ib = fz_wave_incl_scan(pb);
                        const uint32_t ta = __builtin_amdgcn_readlane(ia, 63), tb = __builtin_amdgcn_readlane(ib, 63);
                        const uint32_t tot_s = (ta & 0xffffu) + (tb & 0xffffu), tot_m = (ta >> 16) + (tb >> 16);
                        if (nnext + tot_s > a.cand_cap) { overflow = true; break; }
                        if (tot_m > FZ_GEN_MCAP) {                 // more matches than the buffer holds at once: one slice at a time
                            if (!emit(sa) || !emit(sb)) overflow = true;
                            continue;
                        }
                        if (mb + tot_m > FZ_GEN_MCAP) flush_matches();
                        const uint32_t ea = ia - pa, eb = ib - pb + ta;
                        uint2 *nxa = reinterpret_cast<uint2 *>(nxt) + nnext + (ea & 0xffffu);
                        uint2 *nxb = reinterpret_cast<uint2 *>(nxt) + nnext + (eb & 0xffffu);
                        uint64_t *mpa = mbuf + mb + (ea >> 16), *mpb = mbuf + mb + (eb >> 16);
                        const uint64_t stamp = (uint64_t)index << 48;
                        if (sa.fa) nxa[0] = make_uint2(sa.a0, sa.a1);
                        if (sa.fb) nxa[sa.fa] = make_uint2(sa.b0, sa.b1);
                        if (sa.fc) nxa[sa.fa + sa.fb] = make_uint2(sa.c0, sa.c1);
                        if (sb.fa) nxb[0] = make_uint2(sb.a0, sb.a1);
                        if (sb.fb) nxb[sb.fa] = make_uint2(sb.b0, sb.b1);
                        if (sb.fc) nxb[sb.fa + sb.fb] = make_uint2(sb.c0, sb.c1);
                        if (sa.f1) mpa[0] = (uint64_t)sa.m1 | ((uint64_t)sa.d1 << 32) | stamp;
                        if (sa.f2) mpa[sa.f1] = (uint64_t)sa.m2 | ((uint64_t)sa.d2 << 32) | stamp;
                        if (sb.f1) mpb[0] = (uint64_t)sb.m1 | ((uint64_t)sb.d1 << 32) | stamp;
                        if (sb.f2) mpb[sb.f1] = (uint64_t)sb.m2 | ((uint64_t)sb.d2 << 32) | stamp;
                        nnext += tot_s;
                        mb += tot_m;
                    }
                }
#endif
                for (; c0 < ncur && !overflow; c0 += 64u) {
                    FzGStep st;
                    load_step(c0, st);
                    if (!emit(st)) { overflow = true; break; }
                }
            } else {
                for (uint32_t c0 = 0; c0 < ncur; c0 += 64u) {
                    FzGStep st;
                    fz_gstep_clear(st);
                    if (c0 + lane < ncur) {
                        const uint2 cw = reinterpret_cast<const uint2 *>(cur)[c0 + lane];
                        const FzGCand c = fz_gcand_of(cw.x, cw.y);
                        if (!last) {
                            // slot form (no scratch array); its skip loop has no lane-divergent exit — see fz_device.h for the
                            // hipcc miscompile of the `break` form that kept this out of the kernel in round 4.
                            fz_levlp_step_slots(cw.x, cw.y, ch, index, more_seq, a.m, patf, a.k, st);
                        } else {
                            uint32_t d;
                            const bool hit_end = lev ? fz_levlp_final(c, a.m, a.k, d) : fz_generic_final(c, a.m, a.max_dels, a.k, d);
                            if (hit_end) { st.f1 = 1; st.m1 = (uint32_t)c.start | (wlen << 16); st.d1 = d; }
                        }
                    }
                    if (!emit(st)) { overflow = true; break; }
                }
            }
            // the next character reads what this one stored: a wave's LDS operations are performed in issue order,
            // only the compiler must not reorder; lists in HBM need the real thing
            if (HBM_LISTS) fz_wave_lds_sync(); else asm volatile("" ::: "memory");
            FzGCand *tmp = cur; cur = nxt; nxt = tmp;
            ncur = nnext;
        }
        if (overflow) {
            if (lane == 0) atomicAdd(&counters[2], 1ull);              // host retries with bigger lists
        } else {
            flush_matches();
            if (fold && f_have) emit_pair();
        }
        if (order && q < FZ_GEN_ORDER_MAX && lane == 0) { order_first[q] = 0; order_count[q] = overflow ? 0u : mseq; }
        fz_wave_lds_sync();
    }
    // folded search whose pairs went straight into the host's staging buffer: the last workgroup publishes the counters
    // there as well (no D2H copy command behind the kernel) and leaves them zeroed for the next search
    fz_finish_launch(a, counters, reinterpret_cast<volatile uint32_t *>(smem));
}

// ---- the generic search's automaton, one WORKGROUP of four waves per n-gram hit (round 4) --------------------------
// fz_lp_kernel runs a hit on ONE wave: ~74 window characters one after the other, every character 2-3 slices of 64
// candidates, every slice a dependent chain of ~220 instructions and three LDS round trips — a hit's time is that chain
// (0.15 ms for BASELINE configs[3b], however few hits there are: the kernel lasts as long as its slowest hit, and with
// the window table a third of the waves is left, on a machine that is then mostly idle).  But the candidates of
// different START positions never interact: a candidate's successors keep its start, the list is grouped by start in
// ascending order (fresh candidates are appended, generic_search.py:80), and a match's place in the emission order is
// (window character, start, order among the matches of that start).  So the starts are dealt out to the four waves of
// a workgroup (start mod W, W = 2 or 4 — neighbouring starts carry the large trees of a true occurrence), every wave runs the whole
// window over its own quarter of the list with NO synchronisation, buffers its matches, and at the end the four sorted
// buffers are merged by rank (binary searches on (character, start): no ties across waves).  One slice per character
// instead of 2-3.  A hit whose wave outgrows its list quarter or its match buffer is counted in counters[FZ_HDR_GEN_FAIL]
// and the host runs the search again with fz_lp_kernel (which also keeps the lists-in-HBM form, the file API's segments
// and the tiled whole-sequence automata).
#define FZ_GH_MAX_WAVES 4u                                  // waves per hit: template parameter W in {2, 4}
#define FZ_GH_MCAP_W(W) ((W) == 1u ? 512u : 256u)           // matches a wave buffers for one hit (one wave per hit: the whole hit's)
#define FZ_HDR_GEN_FAIL 5                                   // counters[5]: hits fz_gen_hit_kernel could not finish
#define FZ_GH_CTL_BYTES 64u

__device__ __forceinline__ uint32_t fz_gh_lower_bound(const uint64_t *mb, uint32_t n, uint32_t key) {
    uint32_t lo = 0, hi = n;                                // first entry whose (character << 16 | start) is >= key
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint64_t v = mb[mid];
        const uint32_t kv = ((uint32_t)(v >> 48) << 16) | ((uint32_t)v & 0xffffu);
        if (kv < key) lo = mid + 1u; else hi = mid;
    }
    return lo;
}

// BITS (round 5; patterns of at most 64 characters, budgets of at most 32): the candidate step is fz_generic_step_bits —
// one 64-bit equality word per window character (a 256-entry table of the pattern in LDS, looked up once per window
// character when the window is staged), every flag a 0 / 1 word, the five outputs stored UNCONDITIONALLY (an absent output
// goes to a per-lane dummy slot: no exec-mask juggling around five ds_writes).  With fz_gen_start_useful a window's list
// rarely passes 64 candidates, so this form runs ONE wave per hit (W = 1: no rank merge, no cross-wave traffic).
#define FZ_GH_PT_BYTES 2048u
template <uint32_t W, bool BITS>
__global__ __launch_bounds__(64 * W) void fz_gen_hit_kernel(const uint8_t *__restrict__ buf, const FzScanArgs a,
                                                                      const uint64_t *__restrict__ hits, FzGenRec *__restrict__ recs,
                                                                      unsigned long long *__restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    // (the wave number through v_readfirstlane: the compiler must know it is the same in every lane, or the list lengths
    //  that depend on it — and every loop over them — turn into per-lane values and exec-mask loops)
    const uint32_t wave = fz_uniform(tid >> 6);
    const uint32_t mpad = (a.m + 15u) & ~15u;
    const uint32_t wpad = (a.m + 2u * a.k + 15u) & ~15u;
    const uint32_t capw = fz_uniform(a.cand_cap);           // candidate slots per list of ONE wave
    uint8_t *pat = smem;
    uint8_t *win = smem + mpad;
    volatile uint32_t *ctl = reinterpret_cast<volatile uint32_t *>(smem + mpad + wpad);   // [0] run [1] fail [2..5] matches per wave [6,7] record base [8] stop
    FzGCand *lists = reinterpret_cast<FzGCand *>(smem + mpad + wpad + FZ_GH_CTL_BYTES);
    FzGCand *cur = lists + (size_t)wave * 2u * capw, *nxt = cur + capw;
    uint64_t *mball = reinterpret_cast<uint64_t *>(lists + (size_t)W * 2u * capw);
    constexpr uint32_t FZ_GH_MCAP = FZ_GH_MCAP_W(W);
    uint64_t *mbuf = mball + (size_t)wave * FZ_GH_MCAP;
    // BITS: equality words of the pattern by byte value, of the staged window by position, and the dummy slots
    unsigned long long *ptab = reinterpret_cast<unsigned long long *>(mball + (size_t)W * FZ_GH_MCAP);
    unsigned long long *pwin = ptab + 256;
    uint2 *dummy = reinterpret_cast<uint2 *>(pwin + wpad) + tid;
    fz_copy_pattern(pat, a, tid, 64u * W);
    if constexpr (BITS) {
        for (uint32_t i = tid; i < 256u; i += 64u * W) ptab[i] = 0ull;
        __syncthreads();
        for (uint32_t i = tid; i < a.m; i += 64u * W) atomicOr(&ptab[pat[i]], 1ull << i);
        __syncthreads();
    }
    auto patf = [&](uint32_t i) -> uint8_t { return pat[i]; };
    // (everything that is the same in all lanes goes through v_readfirstlane: values loaded from memory are per-lane values
    //  to the compiler, and list lengths, loop bounds and branch conditions derived from them would become vector registers
    //  and exec-mask loops — the first build of this kernel ran its slice loop that way)
    unsigned long long nitems = fz_bcast64(counters[0]);
    if (nitems > a.hit_cap) nitems = a.hit_cap;
    const bool order = a.gen_order != 0;
    unsigned long long *order_first = reinterpret_cast<unsigned long long *>(a.gen_order);
    uint32_t *order_count = reinterpret_cast<uint32_t *>(order_first + FZ_GEN_ORDER_MAX);
    const bool dedup = a.gen_dedup != 0 && nitems <= FZ_GEN_ORDER_MAX;
    const FzGenDedup dd(a.gen_dedup);
    const bool fold = (a.flags & FZ_FLAG_FOLD) != 0;
    for (uint64_t q = blockIdx.x; q < nitems; q += gridDim.x) {
        __syncthreads();                                    // the previous hit's LDS (window, lists, buffers, ctl) is free
        const uint64_t hit = fz_bcast64(hits[q]);
        const uint32_t s = fz_hit_block(hit) * a.L;
        const uint64_t idx = fz_hit_index(hit);
        const FzSeg sg = fz_segment(a.geom, idx, 0u);
        if (!fz_hit_in_range_s(a, s, idx, sg)) {             // the same for every thread
            if (order && q < FZ_GEN_ORDER_MAX && tid == 0) { order_first[q] = 0; order_count[q] = 0; }
            continue;
        }
        const uint64_t reach = (uint64_t)s + a.k;
        const uint64_t w0 = idx - sg.sa > reach ? idx - reach : sg.sa;               // generic_search.py:231
        uint64_t w1 = idx - s + a.m + a.k;
        if (w1 > sg.se) w1 = sg.se;
        const uint32_t wlen = (uint32_t)(w1 - w0);
        if (tid == 0) {
            uint32_t run = 1u, stop = 0u;
            if ((a.flags & FZ_FLAG_ANY) && __hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) { stop = 1u; run = 0u; }
            if (dedup && !stop) {                           // the window table (fz_device.h: FzGenDedup): only the window's leader runs
                const uint32_t sl = dd.wslot[q];
                if (sl != FZ_GEN_DEDUP_NONE && dd.leader(sl) != (uint32_t)q) {
                    run = 0u;
                    if (order) { order_first[q] = 0; order_count[q] = 0; }
                }
            }
            ctl[0] = run; ctl[1] = 0u; ctl[8] = stop;
        }
        for (uint32_t i = tid; i < wlen; i += 64u * W) {
            const uint8_t c = buf[(w0 - a.geom.buf_off) + i];
            win[i] = c;
            if constexpr (BITS) pwin[i] = ptab[c];
        }
        __syncthreads();
        if (fz_uniform(ctl[8])) break;                      // has_near_match_*: a record exists somewhere
        if (!fz_uniform(ctl[0])) continue;   // a hit of a smaller block runs this window

        // ---- this wave's quarter of the candidate list over the whole window; no synchronisation with the other waves ----
        uint32_t ncur = 0, mb = 0;
        uint32_t pw0l = 0, pw0h = 0, pw1l = 0, pw1h = 0;    // BITS: equality words of window positions lane and 64 + lane (m + 2k <= 128)
        if constexpr (BITS) {
            const unsigned long long e0 = lane < wlen ? pwin[lane] : 0ull, e1 = 64u + lane < wlen ? pwin[64u + lane] : 0ull;
            pw0l = (uint32_t)e0; pw0h = (uint32_t)(e0 >> 32); pw1l = (uint32_t)e1; pw1h = (uint32_t)(e1 >> 32);
        }
        bool fail = false;
        FzGCand *lc = cur, *ln = nxt;
        // end of the window (py:172-177): the survivors that reach the pattern's end by deletions
        auto final_flush = [&]() {
            for (uint32_t c0 = 0; c0 < ncur; c0 += 64u) {
                const bool valid = c0 + lane < ncur;
                const uint2 cw = reinterpret_cast<const uint2 *>(lc)[c0 + lane];
                const FzGCand c = fz_gcand_of(cw.x, cw.y);
                uint32_t d = 0;
                const bool hit_end = valid && fz_generic_final(c, a.m, a.max_dels, a.k, d);
                const unsigned long long mask = __ballot(hit_end);
                const uint32_t tot_m = (uint32_t)__popcll(mask);
                if (mb + tot_m > FZ_GH_MCAP) { fail = true; break; }
                if (hit_end) mbuf[mb + fz_rank(mask)] = (uint64_t)((uint32_t)c.start | (wlen << 16)) | ((uint64_t)d << 32) | ((uint64_t)wlen << 48);
                mb = fz_uniform(mb + tot_m);
            }
        };
        if constexpr (BITS) {
            // The window in two runs with straight-line bodies: characters whose start can still reach the pattern's end
            // (fz_gen_start_useful is monotone: index <= wlen + min(max_dels, max_l) - m) spawn, the others only step what is
            // alive and stop when nothing is.  Round 5's phase stamps (profiles/r05_lab_ab.txt) had shown ~740 of a character's
            // ~1 640 cycles OUTSIDE the trip — loop control: the one loop over "spawn? / last character? / failed? / which half of
            // the equality words?" compiled into a maze of ~15 scalar branches and SGPR-spill reloads per character.
            const uint32_t dlim = a.max_dels < a.k ? a.max_dels : a.k;
            const uint32_t n_spawn = wlen + dlim >= a.m ? (wlen + dlim - a.m + 1u < wlen ? wlen + dlim - a.m + 1u : wlen) : 0u;
            // one window character over this wave's list (the fresh candidate, if any, at list position fresh_at)
            auto one_char = [&](uint32_t index, uint32_t fresh_at) -> bool {
                // (the window's equality words live in two register pairs, lane = window position: v_readlane with the
                //  character's number instead of an LDS round trip at the head of every character's chain; both halves are
                //  read and one is selected — no branch)
                const uint32_t il = index & 63u;
                const uint32_t l0 = (uint32_t)__builtin_amdgcn_readlane((int)pw0l, (int)il), h0 = (uint32_t)__builtin_amdgcn_readlane((int)pw0h, (int)il);
                const uint32_t l1 = (uint32_t)__builtin_amdgcn_readlane((int)pw1l, (int)il), h1 = (uint32_t)__builtin_amdgcn_readlane((int)pw1h, (int)il);
                const bool low = index < 64u;
                const unsigned long long peq = (unsigned long long)(low ? l0 : l1) | ((unsigned long long)(low ? h0 : h1) << 32);
                const uint64_t stamp = (uint64_t)index << 48;
                uint32_t nnext = 0;
                // One TRIP steps U slices of 64 candidates: their loads, steps and prefix scans are independent chains that the
                // hardware overlaps, one pass of offsets, 5 U unconditional stores.  With pruned starts nearly every character
                // is ONE slice: that case is a straight line.
                auto trip = [&](auto uc, uint32_t c0) -> bool {
                    constexpr uint32_t U = decltype(uc)::value;
                    FzGStep st[U];
                    uint32_t packed[U], incl[U], base[U];
#pragma unroll
                    for (uint32_t u = 0; u < U; ++u) {
                        const uint32_t at = c0 + 64u * u + lane;
                        const bool valid = at < ncur;
                        uint2 cw = reinterpret_cast<const uint2 *>(lc)[at];      // (past the list: still this workgroup's LDS)
                        const bool fresh = at == fresh_at;
                        cw.x = valid ? (fresh ? index : cw.x) : 0u;
                        cw.y = valid && !fresh ? cw.y : 0u;
                        fz_generic_step_bits(cw.x, cw.y, peq, index, a.m, a.max_subs, a.max_ins, a.max_dels, a.k, st[u]);
                        const uint32_t vm = valid ? 1u : 0u;
                        st[u].fa &= vm; st[u].fb &= vm; st[u].fc &= vm; st[u].f1 &= vm; st[u].f2 &= vm;
                        packed[u] = (st[u].fa + st[u].fb + st[u].fc) | ((st[u].f1 + st[u].f2) << 16);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < U; ++u) incl[u] = fz_wave_incl_scan(packed[u]);
                    uint32_t tot = 0;
#pragma unroll
                    for (uint32_t u = 0; u < U; ++u) { base[u] = tot; tot += (uint32_t)__builtin_amdgcn_readlane((int)incl[u], 63); }
                    const uint32_t tot_s = tot & 0xffffu, tot_m = tot >> 16;
                    if (nnext + tot_s > capw || mb + tot_m > FZ_GH_MCAP) return false;
#pragma unroll
                    for (uint32_t u = 0; u < U; ++u) {
                        const uint32_t excl = incl[u] - packed[u] + base[u];
                        uint2 *nx = reinterpret_cast<uint2 *>(ln) + nnext + (excl & 0xffffu);
                        uint64_t *mp = mbuf + mb + (excl >> 16);
                        // five stores, none of them conditional: an absent output lands in this lane's dummy slot
                        uint2 *pa = st[u].fa ? nx : dummy;
                        uint2 *pb = st[u].fb ? nx + st[u].fa : dummy;
                        uint2 *pc = st[u].fc ? nx + st[u].fa + st[u].fb : dummy;
                        uint64_t *p1 = st[u].f1 ? mp : reinterpret_cast<uint64_t *>(dummy);
                        uint64_t *p2 = st[u].f2 ? mp + st[u].f1 : reinterpret_cast<uint64_t *>(dummy);
                        *pa = make_uint2(st[u].a0, st[u].a1);
                        *pb = make_uint2(st[u].b0, st[u].b1);
                        *pc = make_uint2(st[u].c0, st[u].c1);
                        *p1 = (uint64_t)st[u].m1 | ((uint64_t)st[u].d1 << 32) | stamp;
                        *p2 = (uint64_t)st[u].m2 | ((uint64_t)st[u].d2 << 32) | stamp;
                    }
                    nnext = fz_uniform(nnext + tot_s);
                    mb = fz_uniform(mb + tot_m);
                    return true;
                };
                bool ok = true;
                if (ncur <= 64u) {
                    ok = trip(std::integral_constant<uint32_t, 1>{}, 0u);
                } else {
                    for (uint32_t c0 = 0; c0 < ncur && ok;) {
                        if (ncur - c0 > 64u) { ok = trip(std::integral_constant<uint32_t, 2>{}, c0); c0 += 128u; }
                        else { ok = trip(std::integral_constant<uint32_t, 1>{}, c0); c0 += 64u; }
                    }
                }
                // the next character reads what this one stored: a wave's LDS operations are performed in issue order
                asm volatile("" ::: "memory");
                FzGCand *tmp = lc; lc = ln; ln = tmp;
                ncur = fz_uniform(nnext);
                return ok;
            };
            uint32_t index = 0;
            for (; index < n_spawn; ++index) {
                uint32_t fresh_at = 0xffffffffu;
                if ((index & (W - 1u)) == wave) {               // this start is ours: the fresh candidate (py:80), taken from registers
                    if (ncur >= capw) { fail = true; break; }
                    fresh_at = ncur;
                    ncur = fz_uniform(ncur + 1u);
                }
                if (ncur != 0u && !one_char(index, fresh_at)) { fail = true; break; }
            }
            if (!fail)
                for (; index < wlen && ncur != 0u; ++index)
                    if (!one_char(index, 0xffffffffu)) { fail = true; break; }
            if (!fail && ncur != 0u) final_flush();
        } else {
        for (uint32_t index = 0; index <= wlen && !fail; ++index) {
            uint32_t nnext = 0;
            // nothing alive and no start left that could still reach the pattern's end: the rest of the window emits nothing
            if (ncur == 0u && !fz_gen_start_useful(index, wlen, a.m, a.max_dels, a.k)) break;
            if (index < wlen) {
                const uint8_t ch = (uint8_t)fz_uniform(win[index]);
                uint32_t fresh_at = 0xffffffffu;
                // this start is ours: the fresh candidate (py:80), taken from registers — unless nothing that starts here can
                // reach the pattern's end inside the window (fz_gen_start_useful)
                if ((index & (W - 1u)) == wave && fz_gen_start_useful(index, wlen, a.m, a.max_dels, a.k)) {
                    if (ncur >= capw) { fail = true; break; }
                    fresh_at = ncur;
                    ncur = fz_uniform(ncur + 1u);
                }
                for (uint32_t c0 = 0; c0 < ncur; c0 += 64u) {
                    const bool valid = c0 + lane < ncur;
                    uint2 cw = reinterpret_cast<const uint2 *>(lc)[c0 + lane];      // (up to 63 slots past the list: still this workgroup's LDS)
                    const bool fresh = c0 + lane == fresh_at;
                    cw.x = valid ? (fresh ? index : cw.x) : 0u;
                    cw.y = valid && !fresh ? cw.y : 0u;
                    FzGStep st;
                    fz_generic_step_packed(cw.x, cw.y, ch, index, a.m, patf, a.max_subs, a.max_ins, a.max_dels, a.k, st);
                    const uint32_t vm = valid ? 1u : 0u;
                    st.fa &= vm; st.fb &= vm; st.fc &= vm; st.f1 &= vm; st.f2 &= vm;
                    const uint32_t packed = (st.fa + st.fb + st.fc) | ((st.f1 + st.f2) << 16);
                    const uint32_t incl = fz_wave_incl_scan(packed);
                    const uint32_t tot = __builtin_amdgcn_readlane(incl, 63);
                    const uint32_t excl = incl - packed;
                    const uint32_t tot_s = tot & 0xffffu, tot_m = tot >> 16;
                    if (nnext + tot_s > capw || mb + tot_m > FZ_GH_MCAP) { fail = true; break; }
                    uint2 *nx = reinterpret_cast<uint2 *>(ln) + nnext + (excl & 0xffffu);
                    uint64_t *mp = mbuf + mb + (excl >> 16);
                    const uint64_t stamp = (uint64_t)index << 48;
                    if (st.fa) nx[0] = make_uint2(st.a0, st.a1);
                    if (st.fb) nx[st.fa] = make_uint2(st.b0, st.b1);
                    if (st.fc) nx[st.fa + st.fb] = make_uint2(st.c0, st.c1);
                    if (st.f1) mp[0] = (uint64_t)st.m1 | ((uint64_t)st.d1 << 32) | stamp;
                    if (st.f2) mp[st.f1] = (uint64_t)st.m2 | ((uint64_t)st.d2 << 32) | stamp;
                    nnext = fz_uniform(nnext + tot_s);
                    mb = fz_uniform(mb + tot_m);
                }
            } else {
                final_flush();
            }
            // the next character reads what this one stored: a wave's LDS operations are performed in issue order
            asm volatile("" ::: "memory");
            FzGCand *tmp = lc; lc = ln; ln = tmp;
            ncur = fz_uniform(nnext);
        }
        }
        if (lane == 0) { ctl[2u + wave] = mb; if (fail) ctl[1] = 1u; }
        __syncthreads();
        if (fz_uniform(ctl[1])) {                           // outgrew a list quarter or a match buffer: the host re-runs with fz_lp_kernel
            if (tid == 0) atomicAdd(&counters[FZ_HDR_GEN_FAIL], 1ull);
            continue;
        }
        uint32_t mbw[W];
        uint32_t total = 0;
#pragma unroll
        for (uint32_t w = 0; w < W; ++w) { mbw[w] = fz_uniform(ctl[2u + w]); total += mbw[w]; }
        if (order && q < FZ_GEN_ORDER_MAX && tid == 0) { order_first[q] = 0; order_count[q] = total; }
        if (total == 0) continue;
        if (fold) {
            // first stage of consolidate_overlapping_matches (common.py:150-159), as in fz_lp_kernel: every match that overlaps
            // the running hull is folded into it; the order of the matches does not matter (best = a total order), so wave 0
            // takes the four buffers one after the other
            if (wave == 0) {
                bool f_have = false;
                uint32_t f_lo = 0, f_hi = 0, f_k1 = 0, f_k2 = 0, f_pairs = 0;
                auto emit_pair = [&]() {
                    if (lane == 0) {
                        // (window table: a pair with an empty hull once per member of the window — see fz_lp_kernel)
                        const uint32_t sl = dedup ? dd.wslot[q] : FZ_GEN_DEDUP_NONE;
                        uint32_t copies = 1u;
                        if (sl != FZ_GEN_DEDUP_NONE && f_lo == f_hi) { copies = dd.nmem[sl]; copies = copies < FZ_GEN_DEDUP_MEMBERS ? copies : FZ_GEN_DEDUP_MEMBERS; }
                        for (uint32_t cpy = 0; cpy < copies; ++cpy) {
                            const unsigned long long slot = atomicAdd(&counters[1], 1ull);
                            if (slot < a.rec_cap) {
                                const uint32_t len = 0xffffu - (f_k1 & 0xffffu);
                                FzGenRec r;
                                r.key = copies > 1u ? hits[dd.mem[sl * FZ_GEN_DEDUP_MEMBERS + cpy]] : hit;
                                r.seq = f_pairs; r.se = f_k2 | ((f_k2 + len) << 16); r.dist = f_k1 >> 16; r.win = f_lo | (f_hi << 16);
                                recs[slot] = r;
                            }
                        }
                    }
                    ++f_pairs;
                    f_have = false;
                };
                auto wave_min = [&](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)fz_wave_incl_min(v), 63); };
                for (uint32_t w = 0; w < W; ++w) {
                    const uint64_t *mbx = mball + (size_t)w * FZ_GH_MCAP;
                    const uint32_t nw = fz_uniform(ctl[2u + w]);
                    for (uint32_t e0 = 0; e0 < nw; e0 += 64u) {
                        const bool have_row = e0 + lane < nw;
                        const uint64_t v = have_row ? mbx[e0 + lane] : 0ull;
                        const uint32_t rs = (uint32_t)v & 0xffffu, re = ((uint32_t)v >> 16) & 0xffffu, rd = (uint32_t)(v >> 32) & 0xffffu;
                        const uint32_t k1 = (rd << 16) | (0xffffu - (re - rs));
                        unsigned long long pending = __ballot(have_row);
                        while (pending) {
                            if (!f_have) {
                                const uint32_t l0 = (uint32_t)__builtin_ctzll(pending);
                                f_lo = (uint32_t)__builtin_amdgcn_readlane((int)rs, (int)l0);
                                f_hi = (uint32_t)__builtin_amdgcn_readlane((int)re, (int)l0);
                                f_k1 = (uint32_t)__builtin_amdgcn_readlane((int)k1, (int)l0);
                                f_k2 = f_lo;
                                f_have = true;
                                pending &= pending - 1ull;
                                continue;
                            }
                            const bool mine = ((pending >> lane) & 1ull) && !(re <= f_lo || rs >= f_hi);
                            const unsigned long long ov = __ballot(mine);
                            if (!ov) { emit_pair(); continue; }
                            const uint32_t mlo = wave_min(mine ? rs : 0xffffffffu), mhi = ~wave_min(mine ? ~re : 0xffffffffu);
                            const uint32_t mk1 = wave_min(mine ? k1 : 0xffffffffu);
                            const uint32_t mk2 = wave_min((mine && k1 == mk1) ? rs : 0xffffffffu);
                            f_lo = mlo < f_lo ? mlo : f_lo;
                            f_hi = mhi > f_hi ? mhi : f_hi;
                            if (mk1 < f_k1 || (mk1 == f_k1 && mk2 < f_k2)) { f_k1 = mk1; f_k2 = mk2; }
                            pending &= ~ov;
                        }
                    }
                }
                if (f_have) emit_pair();
            }
            continue;
        }
        // the hit's records, contiguous and in emission order: record (base + rank), rank = the match's place among the four
        // sorted buffers (its own position + the entries of the other waves with a smaller (character, start))
        if (tid == 0) {
            const unsigned long long base = atomicAdd(&counters[1], (unsigned long long)total);
            ctl[6] = (uint32_t)base; ctl[7] = (uint32_t)(base >> 32);
        }
        __syncthreads();
        const unsigned long long base = (unsigned long long)ctl[6] | ((unsigned long long)ctl[7] << 32);
        for (uint32_t e = lane; e < mb; e += 64u) {
            const uint64_t v = mbuf[e];
            const uint32_t key = ((uint32_t)(v >> 48) << 16) | ((uint32_t)v & 0xffffu);
            uint32_t rank = e;
#pragma unroll
            for (uint32_t w = 0; w < W; ++w)
                if (w != wave) rank += fz_gh_lower_bound(mball + (size_t)w * FZ_GH_MCAP, mbw[w], key);
            if (base + rank < a.rec_cap) {
                FzGenRec r;
                r.key = hit; r.seq = rank; r.se = (uint32_t)v; r.dist = (uint32_t)(v >> 32) & 0xffffu; r.win = order ? (uint32_t)q : sg.j;
                recs[base + rank] = r;
            }
        }
    }
    // folded search whose pairs went straight into the host's staging buffer: the last workgroup publishes the counters
    __syncthreads();
    fz_finish_launch(a, counters, reinterpret_cast<volatile uint32_t *>(smem + mpad + wpad + 60u));
}

// Reference order of the generic search's rows (generic_search.py:221-237: blocks in order, the hits of a block by
// index, the matches of a hit in emission order) restored on the device.  A hit's first row = the rows of all hits
// with a smaller key (block << 56 | index).  Quadratic, tiled 256 x 64 over (hit, other hit) pairs with partial
// sums added atomically (6e3 hits = 2304 tiles); the host orders searches with more than FZ_GEN_ORDER_MAX hits.
__device__ __forceinline__ unsigned long long *counters_rw(const unsigned long long *c) { return const_cast<unsigned long long *>(c); }

__global__ __launch_bounds__(256) void fz_gen_order_kernel(const uint64_t *__restrict__ hits, const FzScanArgs a,
                                                           const unsigned long long *counters) {
    constexpr uint32_t TJ = 64;                                 // other hits per tile: the length of a thread's serial chain
    __shared__ uint64_t skey[TJ];
    __shared__ uint32_t scnt[TJ];
    const unsigned long long n = counters[0];
    if (n > a.hit_cap || n > FZ_GEN_ORDER_MAX || counters[2]) return;
    unsigned long long *first = reinterpret_cast<unsigned long long *>(a.gen_order);
    const uint32_t *count = reinterpret_cast<const uint32_t *>(first + FZ_GEN_ORDER_MAX);
    // rows of hit j: its own, or — window table — those of its window's leader (the hit of the smallest block)
    const bool dedup = a.gen_dedup != 0;
    const FzGenDedup dd(a.gen_dedup);
    auto rows_of = [&](uint32_t j) -> uint32_t {
        if (!dedup) return count[j];
        const uint32_t sl = dd.wslot[j];
        return count[sl == FZ_GEN_DEDUP_NONE ? j : dd.leader(sl)];
    };
    const uint32_t nti = ((uint32_t)n + 255u) / 256u, ntj = ((uint32_t)n + TJ - 1u) / TJ;
    for (uint32_t p = blockIdx.x; p < nti * ntj; p += gridDim.x) {
        const uint32_t i = (p / ntj) * 256u + threadIdx.x, j = (p % ntj) * TJ + threadIdx.x;
        __syncthreads();
        if (threadIdx.x < TJ) {
            skey[threadIdx.x] = j < n ? hits[j] : ~0ull;
            scnt[threadIdx.x] = j < n ? rows_of(j) : 0u;
        }
        __syncthreads();
        if (p % ntj == 0 && threadIdx.x < TJ) {                 // the search's row count: every hit once (tile column 0 of its row)
            uint32_t mine = 0;
            for (uint32_t ii = (p / ntj) * 256u + threadIdx.x; ii < (p / ntj) * 256u + 256u && ii < n; ii += TJ) mine += rows_of(ii);
            if (mine) atomicAdd(&counters_rw(counters)[FZ_HDR_GEN_ROWS], (unsigned long long)mine);
        }
        if (i < n) {
            const uint64_t me = hits[i];
            uint32_t sum = 0;                                   // <= TJ * rows of one hit
#pragma unroll 16
            for (uint32_t t = 0; t < TJ; ++t) sum += skey[t] < me ? scnt[t] : 0u;
            if (sum) atomicAdd(&first[i], (unsigned long long)sum);
        }
    }
}

// ... and every record goes to row (first row of its hit + emission number) as a finished fz_match.
__global__ __launch_bounds__(256) void fz_gen_scatter_kernel(const uint64_t *__restrict__ hits, const FzScanArgs a,
                                                             const FzGenRec *__restrict__ recs, FzOutRow *__restrict__ rows,
                                                             const unsigned long long *__restrict__ counters) {
    const unsigned long long n = counters[0], nr = counters[1];
    if (n > a.hit_cap || n > FZ_GEN_ORDER_MAX || counters[2] || nr > a.rec_cap) return;
    const unsigned long long *first = reinterpret_cast<const unsigned long long *>(a.gen_order);
    const bool dedup = a.gen_dedup != 0;
    const FzGenDedup dd(a.gen_dedup);
    for (unsigned long long r = (unsigned long long)blockIdx.x * 256u + threadIdx.x; r < nr; r += (unsigned long long)gridDim.x * 256u) {
        const FzGenRec rec = recs[r];
        const uint32_t sl = dedup ? dd.wslot[rec.win] : FZ_GEN_DEDUP_NONE;
        if (sl == FZ_GEN_DEDUP_NONE) {                          // a hit on its own
            const unsigned long long at = first[rec.win] + rec.seq;
            if (at < a.rows_cap) rows[at] = fz_gen_row(hits[rec.win], a.L, a.k, 0, rec.se, rec.dist);
            continue;
        }
        if (dd.leader(sl) != rec.win) continue;                 // ran before a hit of a smaller block arrived: that one's records count
        uint32_t nm = dd.nmem[sl];
        nm = nm < FZ_GEN_DEDUP_MEMBERS ? nm : FZ_GEN_DEDUP_MEMBERS;
        for (uint32_t i = 0; i < nm; ++i) {                     // the same match for every hit of the window (the leader is a member too)
            const uint32_t h = dd.mem[sl * FZ_GEN_DEDUP_MEMBERS + i];
            const unsigned long long at = first[h] + rec.seq;
            if (at < a.rows_cap) rows[at] = fz_gen_row(hits[h], a.L, a.k, 0, rec.se, rec.dist);
        }
    }
}

// (f)3  Substitutions-only without the n-gram filter (_find_near_matches_substitutions_lp,
// substitutions_only.py:82-136): every window with Hamming distance <= k.  One lane per start.
__global__ __launch_bounds__(256) void fz_hamming_kernel(const uint8_t *__restrict__ buf, const FzScanArgs a,
                                                         FzRec *__restrict__ recs,
                                                         unsigned long long *__restrict__ counters) {
    __shared__ uint8_t pat_s[FZ_MAX_M];
    const bool in_lds = a.m <= FZ_MAX_M;                               // longer patterns are read from HBM (uniform address: one broadcast load)
    if (in_lds) for (uint32_t i = threadIdx.x; i < a.m; i += blockDim.x) pat_s[i] = a.pat[i];
    __syncthreads();
    const uint8_t *pat_g = reinterpret_cast<const uint8_t *>(a.pat_g);
    const uint64_t lo = a.geom.own_lo;
    uint64_t hi = a.geom.own_hi;                                       // starts i with i + m <= n
    if (a.geom.n < a.m) hi = lo;
    else if (hi > a.geom.n - a.m + 1) hi = a.geom.n - a.m + 1;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t span = hi > lo ? hi - lo : 0;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < span; base += stride) {
        if ((a.flags & FZ_FLAG_ANY) && __hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;   // has_near_match_*
        const uint64_t i = lo + base + threadIdx.x;
        bool ok = base + threadIdx.x < span;
        uint32_t d = 0;
        if (ok) {
            const uint8_t *t = buf + (i - a.geom.buf_off);
            if (in_lds) { for (uint32_t q = 0; q < a.m && d <= a.k; ++q) d += (pat_s[q] != t[q]) ? 1u : 0u; }
            else { for (uint32_t q = 0; q < a.m && d <= a.k; ++q) d += (pat_g[q] != t[q]) ? 1u : 0u; }
            ok = d <= a.k;
        }
        const unsigned long long mask = __ballot(ok);
        if (mask) {
            unsigned long long slot0 = 0;
            if (fz_lane() == 0) slot0 = atomicAdd(&counters[1], (unsigned long long)__popcll(mask));
            slot0 = fz_bcast64(slot0);
            if (ok) {
                const unsigned long long slot = slot0 + fz_rank(mask);
                if (slot < a.rec_cap) { FzRec r; r.key = i; r.l = 0; r.r = 0; r.dist = d; r.aux = 0; recs[slot] = r; }
            }
        }
    }
}
