// fz_device.h — per-candidate verification logic shared by the HIP kernels (fz_kernels.h).
//
// Everything here is FZ_HD (__host__ __device__) so that tests/ can also compile the very same
// functions with g++ and run them lane-by-lane against the oracle in the CPU-only container
// (tests/host_emul.cpp).  The product only ever runs them on the GPU.
//
// Reference semantics (file:line into /root/reference/src/fuzzysearch/), see SURVEY.md App. A:
//   fz_expand           <- c_expand_short / c_expand_long       _levenshtein_ngrams.pyx:9-154
//   fz_verify_lev       <- body of the hit loop                 levenshtein_ngram.py:177-198
//   fz_verify_subs      <- mismatch counting around a hit       _substitutions_only_ngrams_template.h:103-121
//                          + count_differences_with_maximum     common.py:119-142 / substitutions_only.py:266-278
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FZ_HD __host__ __device__ __forceinline__
#else
#define FZ_HD inline
#endif

// "does any lane of the wave want this?" — wave-uniform on the device (a scalar branch), the caller's own flag
// in host code (tests/host_emul.cpp drives one candidate at a time)
#if defined(__HIP_DEVICE_COMPILE__)
#define FZ_WAVE_ANY(x) (__ballot((x) != 0) != 0ull)
#define FZ_WAVE_BALLOT(x) __ballot((x) != 0)
#else
#define FZ_WAVE_ANY(x) ((x) != 0)
#define FZ_WAVE_BALLOT(x) ((x) ? 1ull : 0ull)
#endif

#define FZ_MAX_BLOCKS_PER_LAUNCH 16    // n-gram blocks tested by one filter launch (round 6: 8 -> 16; a pattern of 9 .. 16 blocks is ONE pass)
#define FZ_BLK_BITS 4                  // log2(FZ_MAX_BLOCKS_PER_LAUNCH): bits of the block number in a queue code
#define FZ_MAX_REGIONS 8               // regions of a scan grid (FzScanArgs.reg_*)
#define FZ_MAX_M 1024                  // pattern bytes carried in the kernel argument block (longer patterns: FzScanArgs.pat_g)
#define FZ_MAX_K 255                   // largest budget of the LDS-ring verification and of the candidate automata (8-bit counters)
#define FZ_MAX_M_ANY 65535u            // longest subsequence any path accepts (16-bit window-relative coordinates)
#define FZ_MAX_K_ANY 1023u             // largest Levenshtein / substitution budget (fz_verify_big_kernel: 2k + 1 <= 64 lanes x 32 cells)

// What fz_lp_kernel iterates over.
enum FzLpKind : uint32_t {
    FZ_LP_GENERIC_HIT = 0,   // generic automaton on the window around every n-gram hit (generic_search.py:198-237)
    FZ_LP_GENERIC_SEQ = 1,   // generic automaton over the whole sequence (generic_search.py:57-177), tiled by start
    FZ_LP_LEV_SEQ = 2,       // Levenshtein automaton over the whole sequence (levenshtein.py:52-148), tiled by start
};

enum FzMode : uint32_t { FZ_MODE_EXACT = 0, FZ_MODE_LEV = 1, FZ_MODE_SUBS = 2, FZ_MODE_GENERIC = 3 };

// Geometry of the resident (shard of the) sequence.  All match arithmetic is in GLOBAL
// coordinates; `buf` holds global bytes [buf_off, buf_off + buf_len).
//
// Segments (find_near_matches_in_file, __init__.py:129-200): the reference searches a file chunk by
// chunk, every chunk as an INDEPENDENT sequence (window clamps use the chunk's own ends), so the
// result depends on the chunk geometry.  A whole batch of chunks is resident at
// once here and every chunk is a "segment" with its own clamps:
//     segment j = [seg_org + j*S - pre, seg_org + (j+1)*S + post)  clipped to  [seg_org, n)
// binary files (:129-171): S = chunk_size - keep, pre = 0, post = keep;
// text files   (:174-200): S = chunk_size,        pre = keep, post = 0     (keep = m - 1 + extra).
// seg_stride == 0: one segment, the whole sequence [0, n).  With pre, post <= S / 2 a position lies in
// at most two segments: its core segment (idx - seg_org) / S and the neighbour the extension
// reaches into.  A launch owns the segments [seg_j0, seg_j1).
struct FzGeom {
    uint64_t n;         // global sequence length (every clamp of App. A uses this)
    uint64_t buf_off;   // global index of buf[0]
    uint64_t buf_len;   // valid bytes in buf (the allocation is zero-padded on both sides)
    uint64_t own_lo;    // this shard owns n-gram hits with own_lo <= idx < own_hi
    uint64_t own_hi;
    uint64_t seg_stride;   // S; 0 = unsegmented
    uint64_t seg_org;      // global index where segment 0's core starts
    uint64_t seg_j0, seg_j1;   // segments owned by this launch
    uint32_t seg_pre, seg_post;
};

// One segment's bounds [sa, se).
struct FzSeg { uint64_t sa, se; uint32_t j; uint32_t ok; };

// Candidate segment number c (0 or 1) of position idx; ok = 0 if there is no such segment in this launch.
FZ_HD FzSeg fz_segment(const FzGeom &g, uint64_t idx, uint32_t c) {
    FzSeg r;
    r.j = 0;
    if (g.seg_stride == 0) { r.sa = 0; r.se = g.n; r.ok = c == 0 ? 1u : 0u; return r; }
    r.sa = r.se = 0; r.ok = 0;
    if (idx < g.seg_org) return r;
    uint64_t j = (idx - g.seg_org) / g.seg_stride;
    if (c == 1) {
        if (g.seg_post) { if (j == 0) return r; --j; }       // the previous segment's extension reaches here
        else if (g.seg_pre) ++j;                              // the next segment starts `pre` bytes early
        else return r;
    }
    if (j < g.seg_j0 || j >= g.seg_j1) return r;
    const uint64_t core = g.seg_org + j * g.seg_stride;
    r.sa = (core - g.seg_org >= g.seg_pre) ? core - g.seg_pre : g.seg_org;
    r.se = core + g.seg_stride + g.seg_post;
    if (r.se > g.n) r.se = g.n;
    r.j = (uint32_t)j;
    r.ok = (idx >= r.sa && idx < r.se) ? 1u : 0u;
    return r;
}
FZ_HD uint32_t fz_segment_candidates(const FzGeom &g) { return (g.seg_stride && (g.seg_pre || g.seg_post)) ? 2u : 1u; }

// One scan launch: up to FZ_MAX_BLOCKS_PER_LAUNCH n-gram blocks of length L.  A hit of block b at global index idx is
// accepted for the segment [sa, se) iff
//     sa + lo_rel[b] <= idx  &&  idx + L <= se - hi_sub[b]  &&  abs_lo <= idx  &&  idx + L <= abs_hi
//     &&  own_lo <= idx < own_hi
// (levenshtein_ngram.py:171-176: lo_rel = max(0, s - k), hi_sub = max(0, m - s - L - k);
//  _substitutions_only_ngrams_template.h:97-101: lo_rel = s, hi_sub = m - s - L;
//  search_exact.py:70-71: abs_lo / abs_hi = the clamped start / end index).
struct FzScanArgs {
    FzGeom   geom;
    uint32_t mode;                              // FzMode: what happens to confirmed hits
    uint32_t m, k;                              // pattern length, edit / substitution budget
    uint32_t L;                                 // n-gram length
    uint32_t nblk;                              // real blocks in this launch
    uint32_t g0;                                // global block index of block 0 of this launch
    uint32_t d2;                                // byte offset of the 2nd exact-compare window
    uint32_t mask1;                             // mask of the 1st window ((1 << 8L) - 1 when L < 4)
    uint32_t mask2;                             // mask of the 2nd window (n-gram bytes d2 .. d2+3)
    uint32_t fused;                             // 1: verify inside the scan kernel, 0: emit hits
    uint32_t band_w;                            // rolling score slots per lane (2k + 2)
    uint32_t win_dwords;                        // window dwords staged per lane ((m + 2k + 6) / 4 + 1)
    uint32_t vlanes;                            // lanes of a wave that verify at once (64, 32 or 16): LDS vs lane use
    uint32_t max_subs, max_ins, max_dels;       // generic search limits (k = max_l_dist there)
    uint32_t cand_cap;                          // automaton kernels: candidate slots per list
    uint64_t cand_scratch;                      // 0: both lists in LDS; else device address of per-wave lists in HBM
                                                // (2 * cand_cap slots per workgroup; pathological inputs only)
    uint32_t lp_kind;                           // FzLpKind of fz_lp_kernel
    uint32_t lp_starts;                         // tiled modes: start positions owned by one window
    uint32_t hash_k;                            // odd multiplier of the window hash (24 bits when L > 4)
    uint32_t lut_shift;                         // table slot of a hash h = (h >> lut_shift) & 31
    uint32_t gw;                                // wavefront verification: lanes per candidate (16, 32 or 64)
    uint32_t qcap;                              // fused in-memory scan: queue entries (= prefetched windows) per wave
    uint32_t win_pieces;                        // ... and 16-byte pieces per window (LDS-DMA granule)
    uint32_t flags;                             // tuning knobs (0 in production)
    uint32_t H[FZ_MAX_BLOCKS_PER_LAUNCH];       // fast-path hash of each block's n-gram
    uint32_t A[FZ_MAX_BLOCKS_PER_LAUNCH];       // 1st window value per block (little endian)
    uint32_t B[FZ_MAX_BLOCKS_PER_LAUNCH];       // 2nd window value per block
    uint32_t lo_rel[FZ_MAX_BLOCKS_PER_LAUNCH];  // accepted hit range of a block, relative to the segment ends
    uint32_t hi_sub[FZ_MAX_BLOCKS_PER_LAUNCH];
    uint32_t s[FZ_MAX_BLOCKS_PER_LAUNCH];       // ngram_start of each block inside the pattern
    uint64_t abs_lo, abs_hi;                    // absolute index range (exact search with start / end index)
    uint64_t hit_cap;                           // capacity of the hit list
    uint64_t rec_cap;                           // capacity of the record list
    uint64_t gen_order;                         // generic search, one shard, no segments: device address of the ordering
                                                // area (FZ_GEN_ORDER_MAX x {u64 first row, u32 row count}); 0: the host orders
    uint64_t gen_dedup;                         // generic search: device address of the window table (FzGenDedup layout below);
                                                // 0: every n-gram hit runs the automaton on its own
    uint64_t rows_cap;                          // generic search ordered on the device: rows the row buffer holds
    uint64_t host_hdr;                          // device-visible address of the host copy of the counters
                                                // (0: none); the last workgroup of the launch fills it
    uint64_t pat_g;                             // m > FZ_MAX_M: device address of the pattern (pat[] unused); else 0
    uint8_t  pat[FZ_MAX_M];                     // whole pattern (m <= FZ_MAX_M)
    // Scan grid in regions (nreg = 0: one region, every workgroup strides over all tiles): workgroups [reg_wg0[r],
    // reg_wg0[r] + reg_nwg[r]) stride over the tiles [reg_tile0[r], reg_end[r]) — the last resident round of a launch gets
    // fewer and fewer tiles per workgroup so that the machine does not drain one long workgroup life at a time
    uint32_t nreg;
    uint32_t reg_wg0[FZ_MAX_REGIONS], reg_nwg[FZ_MAX_REGIONS];
    uint64_t reg_tile0[FZ_MAX_REGIONS], reg_end[FZ_MAX_REGIONS];
};

// A hit: (block g << FZ_IDX_BITS) | global idx (48-bit index: sequences below 256 TiB; 16-bit block number).
// A record: what verification produced for one hit.
#define FZ_IDX_BITS 48
#define FZ_MAX_BLOCKS 65535u           // n-gram blocks of one search (block numbers must fit the key's upper 16 bits)
struct FzRec {
    uint64_t key;        // (g << FZ_IDX_BITS) | idx  -> sorting by key restores the reference's order
    uint32_t l;          // bytes consumed to the left of idx   (start = idx - l)
    uint32_t r;          // bytes consumed right of the n-gram  (end   = idx + L + r)
    uint32_t dist;
    uint32_t aux;        // segment number (file API), else 0
};

#define FZ_REC_NONE 0xffffffffu        // FzRec.dist of a slot whose hit did not verify (slot-per-hit kernels)

FZ_HD uint64_t fz_hit_pack(uint32_t g, uint64_t idx) { return ((uint64_t)g << FZ_IDX_BITS) | idx; }
FZ_HD uint32_t fz_hit_block(uint64_t h) { return (uint32_t)(h >> FZ_IDX_BITS); }
FZ_HD uint64_t fz_hit_index(uint64_t h) { return h & ((1ull << FZ_IDX_BITS) - 1ull); }

// Hash of the first min(L, 8) bytes of an n-gram window as the filter's fast path computes it:
//   L > 4:  x  = bytes [0, 4)                  (little-endian dword)
//           yh = low 24 bits of the dword at byte min(L, 8) - 3, i.e. bytes [min(L,8)-3, min(L,8))
//           h  = yh * K + x                    (one v_mad_u32_u24; K < 2^24)
//   L <= 4: h  = (x & mask) * K                (K odd: a bijection of the masked dword)
// Collisions only cost a visit to the exact re-check.
// Six bits of h, (h >> lut_shift) & 63, select the slot of the block-hash table; the host picks K and
// lut_shift per launch so that different block hashes use different slots.
FZ_HD uint32_t fz_hash_windows(uint32_t x, uint32_t yh, uint32_t k) { return (yh & 0xffffffu) * (k & 0xffffffu) + x; }
FZ_HD uint32_t fz_hash_short(uint32_t x_masked, uint32_t k) { return x_masked * k; }

// ---------------------------------------------------------------------------------------------
// Bounded edit-distance expansion == c_expand_short == c_expand_long (SURVEY.md trap 2):
//   D[i][0] = i, D[0][j] = j, D[i][j] = min(D[i-1][j-1] + (sub[i-1] != win[j-1]), D[i][j-1]+1, D[i-1][j]+1)
//   best = len(sub), arg = 0;  for j = 1..len(win): if D[len(sub)][j] <= best: best, arg = D[..][j], j
//   -> (best, arg) if best <= budget else none.
// Evaluated column by column like the reference, but only on the diagonal band |i - j| <= budget:
// a cell outside the band is > budget, and cells > budget can never produce a cell <= budget, so
// the banded result is exact for every outcome that passes `best <= budget` (App. A.1: 0/100000).
// The band of one column is kept in a ring of W = 2*budget + 2 slots per lane (row i lives in slot
// i mod W; LDS on the GPU): Sc.get(slot) / Sc.set(slot, v).
// `sub(i)` / `win(j)` return element i / j of the (possibly reversed) piece and window.
template <class Sc, class SubF, class WinF>
FZ_HD bool fz_expand(Sc &sc, SubF sub, uint32_t sublen, WinF win, uint32_t winlen, uint32_t budget,
                     uint32_t &dist, uint32_t &consumed) {
    if (sublen == 0) { dist = 0; consumed = 0; return true; }       // pyx:28-30
    const uint32_t big = budget + 1;                                // "more than the budget"
    const uint32_t W = 2 * budget + 2;
    uint32_t best = sublen, arg = 0;
    // column 0: D[i][0] = i.  Only rows 1..budget can matter (D[i][0] > budget otherwise).
    {
        uint32_t rows = budget < sublen ? budget : sublen;
        for (uint32_t i = 1; i <= rows; ++i) sc.set(i, i);          // i < W, slot = i
    }
    // columns beyond sublen + budget cannot hold a bottom-row value <= budget
    uint32_t jmax = winlen;
    if (jmax > sublen + budget) jmax = sublen + budget;
    uint32_t i0 = 1, slot0 = 1;                                     // first band row and its slot
    for (uint32_t j = 1; j <= jmax; ++j) {
        const uint8_t ch = win(j - 1);
        // rows in band for this column: i in [max(1, j - budget), min(sublen, j + budget)]
        if (j > budget + 1) { ++i0; slot0 = (slot0 + 1 == W) ? 0 : slot0 + 1; }
        uint32_t i1 = j + budget; if (i1 > sublen) i1 = sublen;
        if (i0 > i1) break;                                         // band left the table (j - budget > sublen)
        uint32_t a, c;
        if (i0 == 1) { a = j - 1; c = j; }                          // D[0][j-1], D[0][j]  (pyx:49-50)
        else { a = sc.get(slot0 == 0 ? W - 1 : slot0 - 1); c = big; }  // D[i0-1][j-1] in band; D[i0-1][j] out
        uint32_t slot = slot0;
        uint32_t cmin = c;                                          // min over this column's band (+ D[0][j])
        for (uint32_t i = i0; i <= i1; ++i) {
            // left neighbour D[i][j-1]: out of band when i - (j-1) > budget -> treat as > budget
            uint32_t b = (i + 1 > j + budget) ? big : sc.get(slot);
            uint32_t v = a + (ch != sub(i - 1) ? 1u : 0u);          // pyx:54-58
            if (b + 1 < v) v = b + 1;
            if (c + 1 < v) v = c + 1;
            sc.set(slot, v);
            c = v;
            a = b;
            if (v < cmin) cmin = v;
            slot = (slot + 1 == W) ? 0 : slot + 1;
        }
        if (i1 == sublen && c <= best) { best = c; arg = j; }       // '<=' -> LAST arg-min (pyx:67-69)
        // column minima never decrease: once every cell is over budget no later column can pass
        // (the reference bails the same way, pyx:61-65)
        if (cmin > budget) break;
    }
    if (best <= budget) { dist = best; consumed = arg; return true; }
    return false;
}

// Register-resident variant of fz_expand for small budgets (K = the search's max_l_dist, <= FZ_REG_BAND_MAX):
// the same DP table, evaluated row by row on the band |j - i| <= K with the 2K+1 cells of a row and
// the 2K+1 window characters they are compared with held in statically indexed registers (the
// loops over the band offset d are fully unrolled) — no LDS round trip per cell, which made the
// ring version latency-bound (~300 cycles per cell).  Same result: the table does not depend on
// the evaluation order, the band K >= budget contains every cell <= budget, the bottom row is
// scanned in ascending j with '<=' (last arg-min) from the column-0 baseline, and a row whose
// minimum exceeds the budget ends the search (row minima never decrease).
#define FZ_REG_BAND_MAX 8
template <int K, class SubF, class WinF>
FZ_HD bool fz_expand_band(SubF sub, uint32_t sublen, WinF win, uint32_t winlen, uint32_t budget,
                          uint32_t &dist, uint32_t &consumed) {
    if (sublen == 0) { dist = 0; consumed = 0; return true; }
    constexpr uint32_t INF = 0x3fffffffu;
    constexpr int W = 2 * K + 1;
    uint32_t cell[W];            // row i: cell[d + K] = D[i][i + d]
    uint32_t chr[W];             // row i: chr[d + K] = win[i + d - 1], the character of column j = i + d
#pragma unroll
    for (int x = 0; x < W; ++x) {
        const int d = x - K;
        cell[x] = (d >= 0 && (uint32_t)d <= winlen) ? (uint32_t)d : INF;      // row 0: D[0][j] = j
        chr[x] = (d >= 0 && (uint32_t)d < winlen) ? (uint32_t)win((uint32_t)d) : 0x100u;   // row 1 compares win[d]
    }
    for (uint32_t i = 1; i <= sublen; ++i) {
        const uint32_t pc = sub(i - 1);
        uint32_t left = INF, rowmin = INF;
#pragma unroll
        for (int x = 0; x < W; ++x) {
            const int d = x - K;
            const int64_t j = (int64_t)i + d;
            const uint32_t diag = cell[x];
            const uint32_t up = (x + 1 < W) ? cell[x + 1] : INF;
            uint32_t v = diag + (chr[x] != pc ? 1u : 0u);
            if (up + 1 < v) v = up + 1;
            if (left + 1 < v) v = left + 1;
            if (j == 0) v = i;                                               // column 0: D[i][0] = i
            if (j < 0 || j > (int64_t)winlen) v = INF;
            cell[x] = v;
            left = v;
            if (v < rowmin) rowmin = v;
        }
        if (rowmin > budget) return false;
#pragma unroll
        for (int x = 0; x + 1 < W; ++x) chr[x] = chr[x + 1];
        const uint32_t nj = i + K;                                           // next row's rightmost column - 1
        chr[W - 1] = nj < winlen ? (uint32_t)win(nj) : 0x100u;
    }
    uint32_t best = sublen, arg = 0;                                         // column-0 baseline (pyx:33-34)
#pragma unroll
    for (int x = 0; x < W; ++x) {
        const int64_t j = (int64_t)sublen + (x - K);
        if (j >= 1 && j <= (int64_t)winlen && cell[x] <= best) { best = cell[x]; arg = (uint32_t)j; }
    }
    if (best <= budget) { dist = best; consumed = arg; return true; }
    return false;
}

// A byte string held in N registers: byte q of the string = byte (q & 3) of r[q >> 2].
// (Round 2 also measured the whole band DP on such registers — four rows per loop trip, every character a static
// byte of a register, dwords streamed a trip ahead: the rows of a flush took 6 300 instead of 11 500 cycles of
// latency but twice the wave instructions, and the fused scan is bound by VALU issue: 0.231 -> 0.235 ms.  Not kept;
// commit a3442c1^..HEAD~ has it with its host test.)
template <int N>
struct FzBytes {
    uint32_t r[N];
    FZ_HD uint32_t at(int q) const { return (r[q >> 2] >> (8 * (q & 3))) & 0xffu; }
};

// Budget-dispatched expansion: register band for k <= MAXK (<= FZ_REG_BAND_MAX), LDS ring otherwise.
// MAXK is a compile-time cap so that a kernel only pays (in VGPRs) for the band widths it may run:
// the fused scan kernel uses 4 (keeps it at <= 64 VGPRs), the stand-alone verify kernel 8.
template <int MAXK, class Sc, class SubF, class WinF>
FZ_HD bool fz_expand_any(Sc &sc, uint32_t k, SubF sub, uint32_t sublen, WinF win, uint32_t winlen, uint32_t budget,
                         uint32_t &dist, uint32_t &consumed) {
    if (k == 1) return fz_expand_band<1>(sub, sublen, win, winlen, budget, dist, consumed);
    if (k == 2) return fz_expand_band<2>(sub, sublen, win, winlen, budget, dist, consumed);
    if (k == 3) return fz_expand_band<3>(sub, sublen, win, winlen, budget, dist, consumed);
    if (k == 4) return fz_expand_band<4>(sub, sublen, win, winlen, budget, dist, consumed);
    if constexpr (MAXK >= 8) {
        if (k == 5) return fz_expand_band<5>(sub, sublen, win, winlen, budget, dist, consumed);
        if (k == 6) return fz_expand_band<6>(sub, sublen, win, winlen, budget, dist, consumed);
        if (k == 7) return fz_expand_band<7>(sub, sublen, win, winlen, budget, dist, consumed);
        if (k == 8) return fz_expand_band<8>(sub, sublen, win, winlen, budget, dist, consumed);
    }
    return fz_expand(sc, sub, sublen, win, winlen, budget, dist, consumed);
}

// Accepted hit range of the block starting at pattern offset s, relative to the ends of the sequence /
// segment: sa + lo_rel <= idx and idx + L <= se - hi_sub.
FZ_HD void fz_block_range(uint32_t mode, uint32_t m, uint32_t k, uint32_t L, uint32_t s, uint32_t &lo_rel, uint32_t &hi_sub) {
    if (mode == FZ_MODE_SUBS) { lo_rel = s; hi_sub = m - s - L; }                  // template.h:97-101
    else if (mode == FZ_MODE_EXACT) { lo_rel = 0; hi_sub = 0; }
    else {                                                                         // levenshtein_ngram.py:171-176
        lo_rel = s > k ? s - k : 0;
        hi_sub = m - s - L > k ? m - s - L - k : 0;
    }
}

// Index-range test of a candidate against one segment (block's accepted hit range, absolute range,
// shard ownership) — the `for index in search_exact(ngram, sequence, start, end)` bounds of
// levenshtein_ngram.py:171-176 / generic_search.py:221-228 / _substitutions_only_ngrams_template.h:97-101.
FZ_HD bool fz_hit_in_range(const FzScanArgs &a, uint32_t blk, uint64_t idx, const FzSeg &sg) {
    if (!sg.ok || blk >= a.nblk) return false;
    if (idx < sg.sa + a.lo_rel[blk]) return false;
    if (sg.se < a.hi_sub[blk] || idx + a.L > sg.se - a.hi_sub[blk]) return false;
    if (idx < a.abs_lo || idx + a.L > a.abs_hi) return false;
    return idx >= a.geom.own_lo && idx < a.geom.own_hi;
}

// The same test for a hit of ANY block of the search (kernels that run after the scan launches).
FZ_HD bool fz_hit_in_range_s(const FzScanArgs &a, uint32_t s, uint64_t idx, const FzSeg &sg) {
    if (!sg.ok) return false;
    uint32_t lo_rel, hi_sub;
    fz_block_range(a.mode, a.m, a.k, a.L, s, lo_rel, hi_sub);
    if (idx < sg.sa + lo_rel) return false;
    if (sg.se < hi_sub || idx + a.L > sg.se - hi_sub) return false;
    if (idx < a.abs_lo || idx + a.L > a.abs_hi) return false;
    return idx >= a.geom.own_lo && idx < a.geom.own_hi;
}

// levenshtein_ngram.py:177-198 for one hit (block starting at s in the pattern, hit at idx) inside the
// sequence / segment [sa, se) (sa = 0, se = n for an in-memory search).
// `t.at(g)` returns the sequence byte at GLOBAL index g; only indices inside
// [max(sa, idx-s-k), min(se, idx-s+m+k)) are ever requested.
template <int MAXK, class Sc, class Seq>
FZ_HD bool fz_verify_lev(Sc &sc, const Seq &t, uint64_t sa, uint64_t se, const uint8_t *p, uint32_t m,
                         uint32_t k, uint32_t L, uint32_t s, uint64_t idx, FzRec &rec) {
    // right: p[s+L:] vs t[idx+L : min(se, idx-s+m+k)]      (idx >= sa+s-k guarantees idx-s+m+k >= sa)
    const uint32_t rlen = m - s - L;
    uint64_t rbeg = idx + L;
    uint64_t rend = idx + m + k - s; if (rend > se) rend = se;
    if (rbeg > se) rbeg = se;
    if (rend < rbeg) rend = rbeg;
    uint32_t dR = 0, r = 0;
    {
        const uint8_t *ps = p + s + L;
        auto sub = [&](uint32_t i) -> uint8_t { return ps[i]; };
        auto win = [&](uint32_t j) -> uint8_t { return t.at(rbeg + j); };
        if (!fz_expand_any<MAXK>(sc, k, sub, rlen, win, (uint32_t)(rend - rbeg), k, dR, r)) return false;
    }
    // left: reversed p[:s] vs reversed t[max(sa, idx-s-(k-dR)) : idx], budget k - dR
    const uint32_t bl = k - dR;
    uint64_t want = (uint64_t)s + bl;
    uint64_t lbeg = (idx - sa > want) ? (idx - want) : sa;
    uint32_t dL = 0, l = 0;
    {
        auto sub = [&](uint32_t i) -> uint8_t { return p[s - 1 - i]; };
        auto win = [&](uint32_t j) -> uint8_t { return t.at(idx - 1 - j); };
        if (!fz_expand_any<MAXK>(sc, k, sub, s, win, (uint32_t)(idx - lbeg), bl, dL, l)) return false;
    }
    rec.l = l; rec.r = r; rec.dist = dL + dR; rec.aux = 0;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Bit-vector expansion — the same table as fz_expand (c_expand_short / c_expand_long, _levenshtein_ngrams.pyx:9-154),
// evaluated one COLUMN (= one sequence character) at a time by the column recurrence of Myers (1999) in Hyyrö's (2001)
// form: a column of the table is held as its vertical differences, two bit vectors over the piece's rows
//     VP bit i: D[i+1][j] - D[i][j] = +1        VN bit i: D[i+1][j] - D[i][j] = -1
// and one step takes the column j-1 to the column j with ~15 word operations, whatever the budget — no band, no cell
// outside it, every cell exact.  The piece's last row is the TOP bit of the vector, so the score D[m'][j] follows from the
// sign bits of the horizontal differences; the reference's `D[0][j] = j` boundary (the window is matched from its first
// character: prefix distance, not the search-anywhere form) is the 1 shifted into the horizontal +1 vector.
// One candidate per LANE; a piece of up to 64 rows is one 64-bit word (NW = 1), up to 128 rows two (NW = 2), up to 32 rows
// half a word (NW = 4 names the 32-bit form: half the vector instructions per column for patterns up to 32 characters).
//
// Both pieces of a hit come out of TWO tables per search, not two per n-gram block: with the whole pattern laid down top-
// aligned, position q at bit q + (64 NW - m) of the forward table Peq[c] and at bit 64 NW - 1 - q of the reversed one,
//     the right piece p[s+L:]         (rows i = q - s - L)  occupies the top  m - s - L  bits of the forward table,
//     the left piece  reversed(p[:s]) (rows i = s - 1 - q)  occupies the top  s          bits of the reversed table
// for EVERY block start s: a piece is selected by masking Eq with its top bits (`hm`), never by a shift.  The rows below
// a piece then hold Eq = 0, VP = 0, VN = 0, stay that way under the recurrence (their horizontal +1 bits are all set, which
// is exactly the boundary's +1 entering the piece's first row) and feed no carry into it.
template <int NW> struct FzBitsWord;
template <> struct FzBitsWord<1> { typedef uint64_t T; };
template <> struct FzBitsWord<2> { typedef unsigned __int128 T; };
template <> struct FzBitsWord<4> { typedef uint32_t T; };
#define FZ_BITS_WIDTH(NW) ((NW) == 4 ? 32u : 64u * (uint32_t)(NW))   // bits of a column vector = longest pattern of the form
#define FZ_BITS_MAX_M(NW) FZ_BITS_WIDTH(NW)

template <int NW> FZ_HD uint32_t fz_bits_fwd_bit(uint32_t m, uint32_t q) { return q + (FZ_BITS_WIDTH(NW) - m); }
template <int NW> FZ_HD uint32_t fz_bits_rev_bit(uint32_t, uint32_t q) { return FZ_BITS_WIDTH(NW) - 1u - q; }

template <int NW>
struct FzBitsCol {
    typename FzBitsWord<NW>::T vp, vn;
};

// a | ~x
template <class T> FZ_HD T fz_or_not(T a, T x) { return a | ~x; }

// Column j-1 -> column j.  `eq`: the piece's rows whose character equals the column's (masked to the piece).
// -> D[m'][j] - D[m'][j-1] (-1, 0 or +1).
template <int NW>
FZ_HD int32_t fz_bits_column(FzBitsCol<NW> &c, typename FzBitsWord<NW>::T eq) {
    typedef typename FzBitsWord<NW>::T T;
    constexpr int TOP = (int)FZ_BITS_WIDTH(NW) - 1;
    const T d0 = (((eq & c.vp) + c.vp) ^ c.vp) | eq | c.vn;     // rows whose diagonal difference is 0
    T hp = fz_or_not<T>(c.vn, d0 | c.vp);                       // horizontal +1
    T hn = c.vp & d0;                                           // horizontal -1
    const int32_t delta = (int32_t)(uint32_t)(uint64_t)(hp >> TOP) - (int32_t)(uint32_t)(uint64_t)(hn >> TOP);
    hp = (hp << 1) | (T)1;                                      // D[0][j] - D[0][j-1] = +1
    hn = hn << 1;
    c.vp = fz_or_not<T>(hn, d0 | hp);
    c.vn = hp & d0;
    return delta;
}

#if defined(__HIP_DEVICE_COMPILE__)
// The one- and two-word columns on the GPU, written on their 32-bit words (only the addition is a wide operation): gfx950
// has a three-input bit operation (v_bitop3_b32) and a funnel shift (v_alignbit_b32), which the compiler only forms from
// 32-bit operands — 36 vector instructions per one-word column instead of 41, 58 instead of 70 per two-word one.  Same
// function, bit for bit (the host model runs the generic one; tests/test_gpu_bits_verify.py holds these against the oracle).
template <int NW>
__device__ __forceinline__ int32_t fz_bits_column_words(FzBitsCol<NW> &c, typename FzBitsWord<NW>::T eq) {
    typedef typename FzBitsWord<NW>::T T;
    constexpr int NH = 2 * NW;                                   // 32-bit words of a vector
    const T sum = (eq & c.vp) + c.vp;
    uint32_t d0[NH], hp[NH], hn[NH];
#pragma unroll
    for (int w = 0; w < NH; ++w) {
        const uint32_t vpw = (uint32_t)(c.vp >> (32 * w)), vnw = (uint32_t)(c.vn >> (32 * w)), eqw = (uint32_t)(eq >> (32 * w));
        d0[w] = (((uint32_t)(sum >> (32 * w)) ^ vpw) | eqw) | vnw;
        hp[w] = vnw | ~(d0[w] | vpw);
        hn[w] = vpw & d0[w];
    }
    const int32_t delta = (int32_t)(hp[NH - 1] >> 31) - (int32_t)(hn[NH - 1] >> 31);
    T nvp = 0, nvn = 0;
#pragma unroll
    for (int w = 0; w < NH; ++w) {
        const uint32_t hps = (hp[w] << 1) | (w ? hp[w ? w - 1 : 0] >> 31 : 1u);      // D[0][j] - D[0][j-1] = +1 enters word 0
        const uint32_t hns = (hn[w] << 1) | (w ? hn[w ? w - 1 : 0] >> 31 : 0u);
        nvp |= (T)(hns | ~(d0[w] | hps)) << (32 * w);
        nvn |= (T)(hps & d0[w]) << (32 * w);
    }
    c.vp = nvp;
    c.vn = nvn;
    return delta;
}
template <> FZ_HD int32_t fz_bits_column<1>(FzBitsCol<1> &c, uint64_t eq) { return fz_bits_column_words<1>(c, eq); }
template <> FZ_HD int32_t fz_bits_column<2>(FzBitsCol<2> &c, unsigned __int128 eq) { return fz_bits_column_words<2>(c, eq); }
#endif

// One expansion on its own (tests, and the statement of what the two-phase loop below computes per side):
// peq(c) = the piece's rows that hold character c, bit i = row i + 1 at bit 64 NW - sublen + i.
template <int NW, class PeqF, class WinF>
FZ_HD bool fz_expand_bits(PeqF peq, uint32_t sublen, WinF win, uint32_t winlen, uint32_t budget,
                          uint32_t &dist, uint32_t &consumed) {
    typedef typename FzBitsWord<NW>::T T;
    if (sublen == 0) { dist = 0; consumed = 0; return true; }       // pyx:28-30
    const T hm = ~(T)0 << (FZ_BITS_WIDTH(NW) - sublen);
    FzBitsCol<NW> c;
    c.vp = hm; c.vn = 0;                                             // column 0: D[i][0] = i
    uint32_t score = sublen, best = sublen, arg = 0;                 // pyx:33-34
    for (uint32_t j = 1; j <= winlen; ++j) {
        score += (uint32_t)fz_bits_column<NW>(c, peq(win(j - 1)) & hm);
        if (score <= best) { best = score; arg = j; }                // '<=' -> LAST arg-min (pyx:67-69)
    }
    if (best <= budget) { dist = best; consumed = arg; return true; }
    return false;
}

// levenshtein_ngram.py:177-198 for one hit by the bit-vector recurrence: right expansion with budget k, then the left one
// with what is left of it, as ONE loop in which every lane walks through its own two expansions — a lane's right piece has
// m - s - L rows and its left one s, so the lanes of a wave differ in both but hardly in the sum (~ m - L + 2k columns):
// run one after the other the wave would pay max(right) + max(left) columns, this way it pays max(right + left).
//   peq.table(side)    -> handle of the forward (0) / reversed (1) table,   peq.at(handle, c) -> its word for character c
//   txt(o)             -> sequence byte at global index wbase + o.  The loop requests a column's Peq word one column
//                         ahead and its character two ahead (on the GPU both are LDS reads: their latency then hides behind
//                         the ~40 instructions of a column), so txt is also asked for up to TWO positions past the end of
//                         an expansion's window (either side); what it returns there is never used.
// `valid` = false: the lane has no candidate (it still takes part in the wave-uniform loop control).
template <int NW, class PeqT, class TxtF>
FZ_HD bool fz_verify_lev_bits(const PeqT &peq, TxtF txt, uint64_t wbase, uint64_t sa, uint64_t se, uint32_t m, uint32_t k,
                              uint32_t L, uint32_t s, uint64_t idx, bool valid, FzRec &rec) {
    typedef typename FzBitsWord<NW>::T T;
    constexpr uint32_t NB = FZ_BITS_WIDTH(NW);
    FzBitsCol<NW> col;
    T hm = 0;
    uint32_t rem = 0, wl = 0, score = 0, key = 0, budget = k, dR = 0, r = 0;
    // start an expansion of a piece of `rows` rows over `cols` window characters: column 0 is D[i][0] = i, the running
    // minimum starts at the column-0 baseline (best = rows, arg = 0: pyx:33-34).  key = (score << 8) | columns left: the
    // minimum over the columns prefers, among equal scores, the LATER column ('<=': last arg-min, pyx:67-69).
    auto start = [&](uint32_t rows, uint32_t cols) {
        hm = rows ? ~(T)0 << (NB - rows) : (T)0;
        col.vp = hm; col.vn = 0;
        score = rows;
        wl = rows ? cols : 0u;                                       // an empty piece expands to (0, 0) (pyx:28-30)
        rem = wl;
        key = (rows << 8) | wl;
    };
    // right: p[s+L:] vs t[idx+L : min(se, idx-s+m+k)]
    uint64_t rbeg = idx + L, rend = idx + m + k - s;
    if (rend > se) rend = se;
    if (rbeg > se) rbeg = se;
    if (rend < rbeg) rend = rbeg;
    start(m - s - L, (uint32_t)(rend - rbeg));
    uint32_t phase = valid ? 0u : 2u;                                // 0 right, 1 left, 2 done
    if (!valid) rem = 0;
    uint32_t pos = (uint32_t)(rbeg - wbase), dir = 1u;
    uint32_t tab = peq.table(0);
    uint32_t chn = 0;                                                // character of the column after the next one
    T eqn = 0;                                                       // Peq word of the next column
    // (the reads of an expansion's first two characters are made whether it has columns or not: txt tolerates them)
    auto prime = [&]() {
        eqn = peq.at(tab, txt(pos));
        chn = txt(pos + dir);
        pos += 2u * dir;
    };
    if (valid) prime();
    bool ok = false;
    uint32_t res_l = 0, res_dist = 0;
    // what the left expansion will be, as far as it does not depend on the right one's outcome
    const T hm_left = s ? ~(T)0 << (NB - s) : (T)0;
    const uint32_t pos_left = (uint32_t)(idx - wbase) - 1u;
    const uint64_t room_left = idx - sa;                             // sequence bytes to the left of the hit
    // Loop control on wave masks (the host model is a wave of one lane): `live` = lanes that still have an expansion to
    // finish.  A column costs ~36 vector instructions; what is added per column for control is one compare and scalar work.
    unsigned long long live = FZ_WAVE_BALLOT(phase < 2u);
    while (live) {
        if (FZ_WAVE_BALLOT(rem == 0u) & live) {                     // some lane's expansion has seen its last column
            if (rem == 0u && phase < 2u) {
                // ONE divergent region with selects inside it, and one WAVE-UNIFORM branch around the left expansion's start
                // (nested divergent branches cost this block more in exec-mask bookkeeping than its arithmetic; most of a
                // wave's ~40 turn events per pass only end lanes — a failed right side, a finished left one — and skip it)
                const uint32_t best = key >> 8, arg = wl - (key & 255u);
                const bool pass = best <= budget;
                const bool to_left = pass && phase == 0u, finish = pass && phase == 1u;
                res_l = finish ? arg : res_l;
                res_dist = finish ? dR + best : res_dist;
                ok = ok || finish;
                phase = to_left ? 1u : 2u;
                if (FZ_WAVE_ANY(to_left)) {
                    // left: reversed p[:s] vs reversed t[max(sa, idx-s-(k-dR)) : idx], budget k - dR
                    dR = to_left ? best : dR;
                    r = to_left ? arg : r;
                    budget = to_left ? k - best : budget;
                    const uint64_t want = (uint64_t)s + budget;
                    const uint32_t lcols = (uint32_t)(room_left > want ? want : room_left);
                    hm = hm_left;
                    col.vp = hm_left; col.vn = 0;
                    score = s;
                    wl = (to_left && s) ? lcols : 0u;                // (an empty piece expands to (0, 0))
                    rem = wl;
                    key = (s << 8) | wl;
                    pos = pos_left; dir = ~0u;
                    tab = peq.table(1);
                    prime();
                }
            }
            live = FZ_WAVE_BALLOT(phase < 2u);
        }
        const uint32_t act = rem != 0u ? 1u : 0u;                    // (rem != 0 only in phases 0 and 1)
        if (act) {
            const T eq = eqn & hm;
            eqn = peq.at(tab, chn);
            chn = txt(pos);
            pos += dir;
            score += (uint32_t)fz_bits_column<NW>(col, eq);
            const uint32_t cand = (score << 8) | (rem - 1u);
            if (cand < key) key = cand;
        }
        rem -= act;
    }
    rec.l = res_l; rec.r = r; rec.dist = res_dist; rec.aux = 0;
    return ok;
}

// _substitutions_only_ngrams_template.h:103-121 for one hit h of the block starting at s:
// window start i = h - s; accept iff Hamming(p, t[i:i+m]) <= k.  The caller guarantees
// s <= h and h - s + m <= n (the block's hit range).  dist = min(Hamming, k+1) (common.py:119-142
// as called from substitutions_only.py:270-274) which, for an accepted window, is the Hamming
// distance itself.
template <class Seq>
FZ_HD bool fz_verify_subs(const Seq &t, const uint8_t *p, uint32_t m, uint32_t k, uint32_t L,
                          uint32_t s, uint64_t h, FzRec &rec) {
    const uint64_t i0 = h - s;
    uint32_t nd = 0;
    for (uint32_t q = 0; q < m; ++q) {
        if (q == s) { q += L - 1; continue; }                       // the n-gram itself matched
        nd += (p[q] != t.at(i0 + q)) ? 1u : 0u;
        if (nd > k) return false;
    }
    rec.l = s; rec.r = m - s - L; rec.dist = nd; rec.aux = 0;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Generic search: the greedy candidate-set automaton of
// _find_near_matches_generic_linear_programming (generic_search.py:57-177, _generic_search.pyx:61-233;
// SURVEY.md trap 6 / App. A.3 — NOT a DP).  One candidate, one window character -> up to three
// successor candidates and up to two matches, in the reference's order.  Coordinates are relative
// to the window (<= m + 2k bytes), so 16 bits suffice.
struct FzGCand {
    uint16_t start, j;                 // window-relative start, next pattern index (subseq_index)
    uint8_t l, ns, ni, nd;             // l_dist, n_subs, n_ins, n_dels
};

struct FzGOut {
    FzGCand succ[3];
    uint32_t nsucc;
    uint32_t mstart[2], mend[2], mdist[2];
    uint32_t nmatch;
};

// Appends with compile-time indices only: a dynamically indexed member array would move the whole
// struct to scratch memory on the GPU (78 scratch accesses in the automaton kernel's inner loop).
FZ_HD void fz_gout_succ(FzGOut &o, const FzGCand &x) {
    if (o.nsucc == 0) o.succ[0] = x;
    else if (o.nsucc == 1) o.succ[1] = x;
    else o.succ[2] = x;
    ++o.nsucc;
}
FZ_HD void fz_gout_match(FzGOut &o, uint32_t start, uint32_t end, uint32_t dist) {
    if (o.nmatch == 0) { o.mstart[0] = start; o.mend[0] = end; o.mdist[0] = dist; }
    else { o.mstart[1] = start; o.mend[1] = end; o.mdist[1] = dist; }
    ++o.nmatch;
}

template <class PatF>
FZ_HD void fz_generic_step(const FzGCand &c, uint8_t ch, uint32_t index, uint32_t m, PatF pat,
                           uint32_t max_subs, uint32_t max_ins, uint32_t max_dels, uint32_t max_l, FzGOut &o) {
    o.nsucc = 0; o.nmatch = 0;
    auto match = [&](uint32_t end, uint32_t dist) { fz_gout_match(o, c.start, end, dist); };
    if (ch == pat(c.j)) {                                              // py:85-94
        if (c.j + 1u == m) match(index + 1, c.l);
        else { FzGCand x = c; x.j = (uint16_t)(c.j + 1); fz_gout_succ(o, x); }
        return;
    }
    if (c.l == max_l) return;                                          // py:101-102
    if (c.ni < max_ins) {                                              // py:104-109: skip a sequence char
        FzGCand x = c; x.ni++; x.l++; fz_gout_succ(o, x);
    }
    if (c.j + 1u < m) {                                                // py:111-128
        if (c.ns < max_subs) {
            FzGCand x = c; x.ns++; x.j++; x.l++; fz_gout_succ(o, x);
        } else if (c.nd < max_dels && c.ni < max_ins) {
            FzGCand x = c; x.ni++; x.nd++; x.j++; x.l++; fz_gout_succ(o, x);
        }
    } else if (c.ns < max_subs || (c.nd < max_dels && c.ni < max_ins)) {   // py:129-138
        match(index + 1, c.l + 1u);
    }
    uint32_t lim = max_dels - c.nd;                                    // py:141-165: skip pattern chars
    if (max_l - c.l < lim) lim = max_l - c.l;
    for (uint32_t sk = 1; sk <= lim; ++sk) {
        if (c.j + sk == m) { match(index, c.l + sk); break; }
        if (pat(c.j + sk) == ch) {
            if (c.j + sk + 1u == m) match(index, c.l + sk);
            else {
                FzGCand x = c; x.nd = (uint8_t)(c.nd + sk); x.j = (uint16_t)(c.j + 1u + sk); x.l = (uint8_t)(c.l + sk);
                fz_gout_succ(o, x);
            }
            break;
        }
    }
}

// The same step on the candidate as it lies in memory — dword 0 = start | j << 16, dword 1 = l | ns << 8 | ni << 16 |
// nd << 24 (FzGCand, little endian) — with the outputs in fixed slots: successor A (advance, or skip a sequence
// character), B (substitution, or insertion + deletion), C (skip pattern characters), match 1 (at index + 1) and
// match 2 (at index), which is the reference's order.  Field updates are additions on the packed words, there are no
// arrays to select into: the kernel's hot form (the struct form above costs ~3x the instructions on the GPU and
// stays as the host-tested statement of the reference; tests/test_device_logic_host.py holds the two together).
struct FzGStep {
    uint32_t a0, a1, b0, b1, c0, c1;
    uint32_t fa, fb, fc;               // 0 / 1: successor present
    uint32_t m1, d1, f1, m2, d2, f2;   // matches: start | end << 16, distance, present
};

FZ_HD void fz_gstep_clear(FzGStep &o) {
    o.a0 = o.a1 = o.b0 = o.b1 = o.c0 = o.c1 = 0; o.fa = o.fb = o.fc = 0;
    o.m1 = o.d1 = o.f1 = o.m2 = o.d2 = o.f2 = 0;
}

// Written without lane-divergent control flow: flags are 0 / 1 words combined with & | ^, the pattern-skip loop
// runs a wave-uniform number of rounds (the profile of the branchy form showed the automaton's waves neither
// executing vector instructions nor waiting for memory for two thirds of their time: exec-mask bookkeeping
// and ~100 branches per 64 candidates).
template <class PatF>
FZ_HD void fz_generic_step_packed(uint32_t w0, uint32_t w1, uint8_t ch, uint32_t index, uint32_t m, PatF pat,
                                  uint32_t max_subs, uint32_t max_ins, uint32_t max_dels, uint32_t max_l, FzGStep &o) {
    // conditions as bools combined with & | ! (never && ||): lane masks in scalar registers on the GPU, where the
    // vector ALU is this kernel's busiest unit
    const uint32_t start = w0 & 0xffffu, j = w0 >> 16;
    const uint32_t l = w1 & 0xffu, ns = (w1 >> 8) & 0xffu, ni = (w1 >> 16) & 0xffu, nd = w1 >> 24;
    // the pattern characters of the first two skip rounds are requested together with pattern[j]: three
    // independent reads instead of a chain of dependent ones (the kernel takes as long as its slowest hit, and a
    // hit's time is this chain, character after character)
    const uint32_t j1 = j + 1u < m ? j + 1u : m - 1u, j2 = j + 2u < m ? j + 2u : m - 1u;
    const uint8_t pj = pat(j);
    const uint8_t p1 = max_dels >= 1u ? pat(j1) : (uint8_t)0, p2 = max_dels >= 2u ? pat(j2) : (uint8_t)0;
    const bool adv = pj == ch;                                         // py:85-94
    const bool at_end = j + 1u == m;
    const bool live = !adv & (l != max_l);                             // py:101-102
    const bool can_ins = ni < max_ins, can_sub = ns < max_subs;
    const bool second = live & (can_sub | ((nd < max_dels) & can_ins));
    o.fa = ((adv & !at_end) | (live & can_ins)) ? 1u : 0u;             // py:104-109
    o.a0 = adv ? w0 + 0x10000u : w0;
    o.a1 = adv ? w1 : w1 + 0x00010001u;                                // ni++, l++
    o.fb = (second & !at_end) ? 1u : 0u;                               // py:111-128
    o.b0 = w0 + 0x10000u;
    o.b1 = w1 + (can_sub ? 0x00000101u : 0x01010001u);                 // ns++, l++  |  ni++, nd++, l++
    o.f1 = ((adv | second) & at_end) ? 1u : 0u;                        // py:86-88, py:129-138
    o.m1 = start | ((index + 1u) << 16);
    o.d1 = adv ? l : l + 1u;
    // py:141-165: the first sk in 1..lim with j + sk == m or pattern[j + sk] == ch
    uint32_t lim = max_dels - nd;
    lim = max_l - l < lim ? max_l - l : lim;
    lim = live ? lim : 0u;
    uint32_t fsk = 0;
    if (max_dels >= 1u) fsk = ((1u <= lim) & ((j + 1u >= m) | (p1 == ch))) ? 1u : 0u;
    if (max_dels >= 2u) fsk = ((2u <= lim) & (fsk == 0u) & ((j + 2u >= m) | (p2 == ch))) ? 2u : fsk;
    for (uint32_t sk = 3; sk <= max_dels; ++sk) {
        const bool open = (sk <= lim) & (fsk == 0u);
        if (!FZ_WAVE_ANY(open ? 1u : 0u)) break;
        const uint32_t pos = j + sk;
        const bool hit = (pos >= m) | (pat(pos < m ? pos : m - 1u) == ch);
        fsk = (open & hit) ? sk : fsk;
    }
    const bool found = fsk != 0u;
    const bool to_end = j + fsk + 1u >= m;                             // ran off the pattern, or matched its last char
    o.f2 = (found & to_end) ? 1u : 0u;
    o.m2 = start | (index << 16);
    o.d2 = l + fsk;
    o.fc = (found & !to_end) ? 1u : 0u;
    o.c0 = w0 + ((1u + fsk) << 16);
    o.c1 = w1 + ((fsk << 24) | fsk);
}

// The same step again for patterns of at most 64 characters and budgets of at most 32 (round 5): the comparisons
// pattern[j] == ch, pattern[j + 1] == ch, ... are bits of ONE 64-bit word per sequence character — peq, bit i set iff
// pattern[i] == ch (a 256-entry table per pattern, one look-up per window character, the same for every candidate) — and
// the skip search of py:141-165 ("the first sk with j + sk == m or pattern[j + sk] == ch") is one find-first-set on
// (peq with a sentinel bit at m) >> (j + 1).  Every flag is a 0 / 1 WORD computed with integer arithmetic — no
// comparison results in scalar registers, no lane-mask algebra on the scalar unit, no pattern reads from LDS, no loop:
// fz_gen_hit_kernel's time is the dependent instruction chain of this step, once per window character (round 4 measured
// ~2 300 cycles per character for the form above: ~150 instructions of which every second hops between the vector and
// the scalar unit).  Same outputs as fz_generic_step_packed, bit for bit (tests/host_emul.cpp holds them together).
FZ_HD uint32_t fz_b_lt(uint32_t a, uint32_t b) { return (a - b) >> 31; }                  // a < b for values below 2^31
FZ_HD uint32_t fz_b_eq(uint32_t a, uint32_t b) { return ((a ^ b) - 1u) >> 31; }           // a == b (a ^ b below 2^31)

FZ_HD void fz_generic_step_bits(uint32_t w0, uint32_t w1, uint64_t peq, uint32_t index, uint32_t m,
                                uint32_t max_subs, uint32_t max_ins, uint32_t max_dels, uint32_t max_l, FzGStep &o) {
    const uint32_t start = w0 & 0xffffu, j = w0 >> 16;
    const uint32_t l = w1 & 0xffu, ns = (w1 >> 8) & 0xffu, ni = (w1 >> 16) & 0xffu, nd = w1 >> 24;
    const uint64_t skipw = (peq >> 1) | (1ull << (m - 1u));             // bit (j + sk - 1): j + sk == m or pattern[j + sk] == ch
    const uint32_t adv = (uint32_t)(peq >> j) & 1u;                     // py:85-94
    const uint32_t skips = (uint32_t)(skipw >> j);                      // bit sk - 1 for sk = 1 .. 32
    const uint32_t nadv = adv ^ 1u;
    const uint32_t at_end = fz_b_eq(j + 1u, m), not_end = at_end ^ 1u;
    const uint32_t live = nadv & (fz_b_eq(l, max_l) ^ 1u);              // py:101-102
    const uint32_t can_ins = fz_b_lt(ni, max_ins), can_sub = fz_b_lt(ns, max_subs);
    const uint32_t second = live & (can_sub | (fz_b_lt(nd, max_dels) & can_ins));
    o.fa = (adv & not_end) | (live & can_ins);                          // py:104-109
    o.a0 = w0 + (adv << 16);
    o.a1 = w1 + nadv * 0x00010001u;                                     // ni++, l++
    o.fb = second & not_end;                                            // py:111-128
    o.b0 = w0 + 0x10000u;
    o.b1 = w1 + 0x01010001u - can_sub * (0x01010001u - 0x00000101u);    // ns++, l++  |  ni++, nd++, l++
    o.f1 = (adv | second) & at_end;                                     // py:86-88, py:129-138
    o.m1 = start | ((index + 1u) << 16);
    o.d1 = l + nadv;
    uint32_t lim = max_dels - nd;                                       // py:141-165
    const uint32_t lim2 = max_l - l;
    lim = lim2 < lim ? lim2 : lim;
    lim *= live;
    uint32_t fsk = (uint32_t)__builtin_ffs((int)skips);                 // 0: no such sk among the first 32
    fsk *= fz_b_lt(fsk, lim + 1u);                                      // beyond the budget: none
    const uint32_t found = fz_b_lt(0u, fsk);
    const uint32_t to_end = fz_b_lt(m, j + fsk + 2u);                   // j + fsk + 1 >= m: ran off the pattern, or matched its last char
    o.f2 = found & to_end;
    o.m2 = start | (index << 16);
    o.d2 = l + fsk;
    o.fc = found & (to_end ^ 1u);
    o.c0 = w0 + ((1u + fsk) << 16);
    o.c1 = w1 + ((fsk << 24) | fsk);
}

FZ_HD void fz_gcand_words(const FzGCand &c, uint32_t &w0, uint32_t &w1) {
    w0 = (uint32_t)c.start | ((uint32_t)c.j << 16);
    w1 = (uint32_t)c.l | ((uint32_t)c.ns << 8) | ((uint32_t)c.ni << 16) | ((uint32_t)c.nd << 24);
}
FZ_HD FzGCand fz_gcand_of(uint32_t w0, uint32_t w1) {
    FzGCand c;
    c.start = (uint16_t)w0; c.j = (uint16_t)(w0 >> 16);
    c.l = (uint8_t)w1; c.ns = (uint8_t)(w1 >> 8); c.ni = (uint8_t)(w1 >> 16); c.nd = (uint8_t)(w1 >> 24);
    return c;
}
// struct form -> slot form (the tiled Levenshtein automaton and the end-of-window flush: not hot)
FZ_HD void fz_gstep_from_out(const FzGOut &g, FzGStep &o) {
    fz_gstep_clear(o);
    if (g.nsucc > 0) { o.fa = 1; fz_gcand_words(g.succ[0], o.a0, o.a1); }
    if (g.nsucc > 1) { o.fb = 1; fz_gcand_words(g.succ[1], o.b0, o.b1); }
    if (g.nsucc > 2) { o.fc = 1; fz_gcand_words(g.succ[2], o.c0, o.c1); }
    if (g.nmatch > 0) { o.f1 = 1; o.m1 = g.mstart[0] | (g.mend[0] << 16); o.d1 = g.mdist[0]; }
    if (g.nmatch > 1) { o.f2 = 1; o.m2 = g.mstart[1] | (g.mend[1] << 16); o.d2 = g.mdist[1]; }
}

// find_near_matches_levenshtein_linear_programming (levenshtein.py:52-148): one candidate, one
// sequence character.  `more_seq` is the reference's `index + 1 < len(sequence)` (global).
template <class PatF>
FZ_HD void fz_levlp_step(const FzGCand &c, uint8_t ch, uint32_t index, bool more_seq, uint32_t m, PatF pat,
                         uint32_t k, FzGOut &o) {
    o.nsucc = 0; o.nmatch = 0;
    auto match = [&](uint32_t end, uint32_t dist) { fz_gout_match(o, c.start, end, dist); };
    if (pat(c.j) == ch) {                                              // :84-92
        if (c.j + 1u == m) match(index + 1, c.l);
        else { FzGCand x = c; x.j = (uint16_t)(c.j + 1); fz_gout_succ(o, x); }
        return;
    }
    if (c.l == k) return;                                              // :99-100
    { FzGCand x = c; x.l++; fz_gout_succ(o, x); }                   // :103 skip a sequence char
    if (more_seq && c.j + 1u < m) {                                    // :105-111 skip both
        FzGCand x = c; x.l++; x.j++; fz_gout_succ(o, x);
    }
    for (uint32_t sk = 1; sk <= k - c.l; ++sk) {                       // :114-137 skip pattern chars
        if (c.j + sk == m) { match(index + 1, c.l + sk); break; }
        if (pat(c.j + sk) == ch) {
            if (c.j + sk + 1u == m) match(index + 1, c.l + sk);
            else { FzGCand x = c; x.l = (uint8_t)(c.l + sk); x.j = (uint16_t)(c.j + 1u + sk); fz_gout_succ(o, x); }
            break;
        }
    }
}

// The same step with its outputs in the fixed slots of FzGStep (what fz_lp_kernel stores through wave prefix sums):
// successors in list order a (:103), b (:105-111), c (:114-137), at most one match.  No arrays: the struct form's
// succ[] was indexed dynamically and lived in scratch memory (20 bytes per lane in the tiled Levenshtein automaton).
//
// The pattern-skip loop (:114-137) is written WITHOUT a lane-divergent exit: every lane runs the same rounds and keeps
// the first sk that ends its search in `fsk`; the loop leaves on a wave-uniform condition only (as the generic step
// above does).  Round 4 had the reference's shape here — `for (sk ..) { if (hit) { ..; break; } }` — and hipcc
// (ROCm 7.2.0, gfx950, -O2 / -O3, inlined into fz_lp_kernel) miscompiled it: the VGPR holding successor a's first word
// (live across the loop) was reused for the temporary j + sk + 1 of the inner test and restored on ONE side of the
// branch that follows, so lanes whose skip ended on the last pattern character stored the successor (start = m, j = 0)
// instead of (start, j): one lost match per ~25 and ghost matches at tile offset + m.  Root-caused in round 5 (ISA in
// profiles/r05_levlp_miscompile.txt; -O1, noinline or this loop shape compile correctly; extra wave syncs change
// nothing — it never was a race).  benchmarks/lab_build.sh x -DFZ_LAB_ONLY -DFZ_LEVLP_BREAKLOOP (applies benchmarks/lab_patches/) + benchmarks/repro_lp.py
// rebuilds and shows the miscompiled form.
template <class PatF>
FZ_HD void fz_levlp_step_slots(uint32_t w0, uint32_t w1, uint8_t ch, uint32_t index, bool more_seq, uint32_t m, PatF pat,
                               uint32_t k, FzGStep &o) {
    fz_gstep_clear(o);
    const uint32_t start = w0 & 0xffffu, j = w0 >> 16, l = w1 & 0xffu;
    if (pat(j) == ch) {                                                // :84-92
        if (j + 1u == m) { o.f1 = 1; o.m1 = start | ((index + 1u) << 16); o.d1 = l; }
        else { o.fa = 1; o.a0 = w0 + 0x10000u; o.a1 = w1; }
        return;
    }
    if (l == k) return;                                                // :99-100
    o.fa = 1; o.a0 = w0; o.a1 = w1 + 1u;                               // :103 skip a sequence char (l++)
    if (more_seq && j + 1u < m) { o.fb = 1; o.b0 = w0 + 0x10000u; o.b1 = w1 + 1u; }   // :105-111 skip both
    uint32_t fsk = 0;                                                  // :114-137: the first sk in 1..k-l with j + sk == m or pattern[j + sk] == ch
    for (uint32_t sk = 1; sk <= k; ++sk) {
        const bool open = (sk <= k - l) & (fsk == 0u);
        if (!FZ_WAVE_ANY(open ? 1u : 0u)) break;                       // wave-uniform: no lane is still looking
        const uint32_t pos = j + sk;
        const bool hit = (pos >= m) | (pat(pos < m ? pos : m - 1u) == ch);
        fsk = (open & hit) ? sk : fsk;
    }
    if (fsk != 0u) {
        if (j + fsk + 1u >= m) { o.f1 = 1; o.m1 = start | ((index + 1u) << 16); o.d1 = l + fsk; }   // ran off the pattern, or matched its last char
        else { o.fc = 1; o.c0 = w0 + ((1u + fsk) << 16); o.c1 = w1 + fsk; }
    }
}

// levenshtein.py:144-148
FZ_HD bool fz_levlp_final(const FzGCand &c, uint32_t m, uint32_t k, uint32_t &dist) {
    dist = c.l + m - c.j;
    return dist <= k;
}

// Which window positions need a fresh candidate at all (round 5).  The automaton of a hit runs over the window
// [idx - s - k, idx - s + m + k) (generic_search.py:229-231) and the reference spawns a candidate at EVERY window character
// (py:80).  A candidate that starts at `index` consumes at most wlen - index characters; j advances by at most one per
// consumed character plus the pattern characters it deletes (nd <= max_dels, and nd <= l <= max_l), and every way of
// emitting a match — the last pattern character consumed (py:86-88, 129-138), skipping to the pattern's end (py:141-165),
// the end-of-window flush (py:172-177) — needs j + remaining deletions >= m.  So a start with
//     index + m > wlen + min(max_dels, max_l)
// can never emit anything, and since candidates of different starts never interact (successors keep their start, the list
// only grows by appending) leaving it out changes neither the matches nor their order.  BASELINE configs[3b] (m = 64,
// k = 5, max_dels = 2): 13 of a window's 74 characters spawn.
FZ_HD bool fz_gen_start_useful(uint32_t index, uint32_t wlen, uint32_t m, uint32_t max_dels, uint32_t max_l) {
    const uint32_t d = max_dels < max_l ? max_dels : max_l;
    return index + m <= wlen + d;
}

// End-of-window flush of one surviving candidate (py:172-177): -> true and dist if it matches.
FZ_HD bool fz_generic_final(const FzGCand &c, uint32_t m, uint32_t max_dels, uint32_t max_l, uint32_t &dist) {
    const uint32_t sk = m - c.j;
    if (c.nd + sk <= max_dels && c.l + sk <= max_l) { dist = c.l + sk; return true; }
    return false;
}

// Record of the generic search: one emitted match of the automaton run on the window of hit `key`.
struct FzGenRec {
    uint64_t key;        // per-hit mode: (block << FZ_IDX_BITS) | idx of the n-gram hit; tiled modes: global step index
    uint32_t seq;        // emission number within the work item (hit window / tile)
    uint32_t se;         // window-relative start | end << 16
    uint32_t dist;
    uint32_t win;        // tiled modes: window (tile) number; per-hit mode: segment number, or (no segments) the
                         // hit's slot in the hit list
};

// Window table of the generic search (round 4).  The automaton's input is the window [idx - s - k, idx - s + m + k) of
// an n-gram hit (generic_search.py:229-231): it depends on idx - s only, and the hits that the different n-gram blocks of
// ONE occurrence produce share it (a planted copy with e edits is found by up to G - e blocks: BASELINE configs[3b] has
// 3x as many hits as distinct windows).  The reference runs its automaton once per hit and emits the same matches each
// time; here the scan enters every hit it lists into a table of windows (fz_gen_claim: the window's slot by atomicCAS, the
// hit as a member, the window's LEADER = its hit of the smallest block by atomicMax on the inverted (block, slot) word),
// and the automaton kernel runs the leaders only:
//   ordered form: fz_gen_order_kernel counts the leader's rows for every member and fz_gen_scatter_kernel writes every row
//     of the leader once per member, with the member's block number;
//   folded / flag-only form: the other hits emit nothing (a consolidation of equal matches keeps the smallest block).
// Layout at FzScanArgs.gen_dedup: u64 keys[T] (idx + k - s + 1; 0 = free), u64 best[T] (~((block << 32) | hit-list slot),
// maximum = smallest block), u32 nmem[T], u32 mem[T][FZ_GEN_DEDUP_MEMBERS], u32 wslot[FZ_GEN_ORDER_MAX] (table slot of
// every hit, or FZ_GEN_DEDUP_NONE: the hit keeps its own rows).  keys, best and nmem must be zero when a search starts.
#define FZ_GEN_DEDUP_SLOTS 32768u
#define FZ_GEN_DEDUP_MEMBERS 7u
#define FZ_GEN_DEDUP_NONE 0xffffffffu
#define FZ_GEN_DEDUP_ZERO_BYTES ((size_t)FZ_GEN_DEDUP_SLOTS * 20u)
#define FZ_GEN_DEDUP_BYTES ((size_t)FZ_GEN_DEDUP_SLOTS * (8u + 8u + 4u + 4u * FZ_GEN_DEDUP_MEMBERS) + (size_t)16384u * 4u)
#define FZ_HDR_GEN_ROWS 4                              // counters[4]: rows of the ordered generic search (hits x their window's matches)

struct FzGenDedup {
    unsigned long long *keys, *best;
    uint32_t *nmem, *mem, *wslot;
    FZ_HD explicit FzGenDedup(uint64_t base) {
        keys = reinterpret_cast<unsigned long long *>(base);
        best = keys + FZ_GEN_DEDUP_SLOTS;
        nmem = reinterpret_cast<uint32_t *>(best + FZ_GEN_DEDUP_SLOTS);
        mem = nmem + FZ_GEN_DEDUP_SLOTS;
        wslot = mem + (size_t)FZ_GEN_DEDUP_SLOTS * FZ_GEN_DEDUP_MEMBERS;
    }
    FZ_HD uint32_t leader(uint32_t slot) const { return (uint32_t)~best[slot]; }     // hit-list slot of the smallest block's hit
};

// The generic search's records are ordered on the device when one shard without segments produced at most this
// many n-gram hits (fz_gen_order_kernel is quadratic in the hit count: 6e3 hits = 10 us); the host orders the rest.
#define FZ_GEN_ORDER_MAX 16384u

// One raw match as the C-ABI returns it (fz_match of include/fzhip.h; fzhip.hip asserts the layout).
struct FzOutRow { int64_t start, end; int32_t dist, block; };

// generic_search.py:229-237: the match `se` (window-relative start | end << 16) of the automaton run on the window
// of n-gram hit `key`, in sequence coordinates.  `sa` = start of the hit's segment (0 without segments).
FZ_HD FzOutRow fz_gen_row(uint64_t key, uint32_t L, uint32_t k, uint64_t sa, uint32_t se, uint32_t dist) {
    const uint64_t idx = fz_hit_index(key);
    const uint32_t blk = fz_hit_block(key);
    const uint64_t reach = (uint64_t)blk * L + k;
    const uint64_t w0 = idx - sa > reach ? idx - reach : sa;
    FzOutRow r;
    r.start = (int64_t)(w0 + (se & 0xffffu));
    r.end = (int64_t)(w0 + (se >> 16));
    r.dist = (int32_t)dist;
    r.block = (int32_t)blk;
    return r;
}

// Plain byte accessor over a resident buffer in GLOBAL coordinates (host emulation / tests).
struct FzSeqView {
    const uint8_t *buf;
    uint64_t buf_off;
    FZ_HD uint8_t at(uint64_t gidx) const { return buf[gidx - buf_off]; }
};
