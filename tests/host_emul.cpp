// host_emul.cpp — TEST HARNESS: compiles the FZ_HD verification functions of
// fuzzysearch_amd/csrc/fz_device.h with g++ and drives them one "lane" at a time, so the banded
// expansion / clamping logic that the GPU kernels run can be checked against the oracle in the
// CPU-only build container.  The n-gram hit enumeration here is a plain memcmp loop standing in
// for the filter kernel (which only exists as HIP and is tested with -m gpu).
// Never part of the product: built and loaded by tests/test_device_logic_host.py only.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../fuzzysearch_amd/csrc/fz_device.h"

struct HostScores {
    std::vector<uint16_t> v;
    uint32_t get(uint32_t i) const { return v[i]; }
    void set(uint32_t i, uint32_t x) { v[i] = (uint16_t)x; }
};

struct OutRec { int64_t start, end; int32_t dist, block; };

extern "C" {

// mode 1 = Levenshtein n-grams, 2 = substitutions-only n-grams.  The sequence is presented to the
// device functions as a shard buffer [buf_off, buf_off + buf_len) of a global sequence of n bytes
// (t points at global byte 0) owning hits in [own_lo, own_hi).
int64_t emul_search(int mode, const uint8_t *p, uint32_t m, const uint8_t *t, uint64_t n, uint32_t k,
                    uint64_t buf_off, uint64_t buf_len, uint64_t own_lo, uint64_t own_hi,
                    OutRec *out, int64_t cap) {
    const uint32_t L = m / (k + 1);
    if (L == 0) return -1;
    HostScores sc;
    sc.v.assign(2 * k + 4, 0);
    // copy the shard so that any out-of-shard access reads poison, not neighbouring data
    std::vector<uint8_t> shard(buf_len + 64, 0xEE);
    memcpy(shard.data(), t + buf_off, buf_len);
    FzSeqView view{shard.data(), buf_off};
    int64_t cnt = 0;
    uint32_t g = 0;
    for (uint32_t s = 0; s + L <= m; s += L, ++g) {
        int64_t lo, hi;
        const int64_t N = (int64_t)n;
        if (mode == 1) {
            lo = (int64_t)s - (int64_t)k; if (lo < 0) lo = 0; if (lo > N) lo = N;
            hi = N - (int64_t)m + s + L + k; if (hi > N) hi = N; if (hi < lo) hi = lo;
        } else {
            if (n < m) return 0;
            lo = s; hi = N - (int64_t)(m - s - L);
        }
        for (int64_t idx = lo; idx + (int64_t)L <= hi; ++idx) {
            if ((uint64_t)idx < own_lo || (uint64_t)idx >= own_hi) continue;
            if (memcmp(t + idx, p + s, L) != 0) continue;
            FzRec rec;
            bool ok = mode == 1 ? fz_verify_lev<FZ_REG_BAND_MAX>(sc, view, 0, n, p, m, k, L, s, (uint64_t)idx, rec)
                                : fz_verify_subs(view, p, m, k, L, s, (uint64_t)idx, rec);
            if (!ok) continue;
            if (cnt < cap) {
                out[cnt].start = idx - (int64_t)rec.l;
                out[cnt].end = idx + L + rec.r;
                out[cnt].dist = (int32_t)rec.dist;
                out[cnt].block = (int32_t)g;
            }
            ++cnt;
        }
    }
    return cnt;
}

// Levenshtein n-gram search over a batch of file chunks ("segments", FzGeom): every n-gram occurrence in the
// resident buffer [buf_off, buf_off + buf_len) is offered to each candidate segment exactly as the scan
// kernel's flush does (fz_segment -> fz_hit_in_range -> fz_verify_lev with the segment's ends).
struct OutRecSeg { int64_t start, end; int32_t dist, block; int64_t seg; };
int64_t emul_search_segments(const uint8_t *p, uint32_t m, const uint8_t *t, uint64_t n, uint32_t k,
                             uint64_t S, uint32_t pre, uint32_t post, uint64_t j0, uint64_t j1,
                             uint64_t buf_off, uint64_t buf_len, OutRecSeg *out, int64_t cap) {
    const uint32_t L = m / (k + 1);
    if (L == 0) return -1;
    HostScores sc;
    sc.v.assign(2 * k + 4, 0);
    std::vector<uint8_t> shard(buf_len + 64, 0xEE);
    memcpy(shard.data(), t + buf_off, buf_len);
    FzSeqView view{shard.data(), buf_off};
    FzScanArgs a;
    memset(&a, 0, sizeof a);
    a.geom.n = n; a.geom.buf_off = buf_off; a.geom.buf_len = buf_len; a.geom.own_lo = 0; a.geom.own_hi = n;
    a.geom.seg_stride = S; a.geom.seg_org = 0; a.geom.seg_pre = pre; a.geom.seg_post = post;
    a.geom.seg_j0 = j0; a.geom.seg_j1 = j1;
    a.mode = FZ_MODE_LEV; a.m = m; a.k = k; a.L = L; a.abs_lo = 0; a.abs_hi = ~0ull;
    int64_t cnt = 0;
    uint32_t g = 0;
    for (uint32_t s = 0; s + L <= m; s += L, ++g) {
        for (uint64_t idx = buf_off; idx + L <= buf_off + buf_len; ++idx) {
            if (memcmp(shard.data() + (idx - buf_off), p + s, L) != 0) continue;
            for (uint32_t c = 0; c < fz_segment_candidates(a.geom); ++c) {
                const FzSeg sg = fz_segment(a.geom, idx, c);
                if (!fz_hit_in_range_s(a, s, idx, sg)) continue;
                FzRec rec;
                if (!fz_verify_lev<FZ_REG_BAND_MAX>(sc, view, sg.sa, sg.se, p, m, k, L, s, idx, rec)) continue;
                if (cnt < cap) {
                    out[cnt].start = (int64_t)idx - (int64_t)rec.l;
                    out[cnt].end = (int64_t)idx + L + rec.r;
                    out[cnt].dist = (int32_t)rec.dist;
                    out[cnt].block = (int32_t)g;
                    out[cnt].seg = sg.j;
                }
                ++cnt;
            }
        }
    }
    return cnt;
}

// Generic automaton over one window, driven candidate by candidate through fz_generic_step exactly
// as the GPU kernel does (the kernel only parallelises the candidate loop and keeps the order).
int64_t emul_generic_lp(const uint8_t *p, uint32_t m, const uint8_t *t, uint32_t n, uint32_t max_subs,
                        uint32_t max_ins, uint32_t max_dels, uint32_t max_l, OutRec *out, int64_t cap) {
    std::vector<FzGCand> cur, nxt;
    int64_t cnt = 0;
    auto pat = [&](uint32_t i) -> uint8_t { return p[i]; };
    auto emit = [&](uint32_t s, uint32_t e, uint32_t d) {
        if (cnt < cap) { out[cnt].start = s; out[cnt].end = e; out[cnt].dist = (int32_t)d; out[cnt].block = -1; }
        ++cnt;
    };
    for (uint32_t index = 0; index < n; ++index) {
        FzGCand fresh{(uint16_t)index, 0, 0, 0, 0, 0};
        cur.push_back(fresh);
        nxt.clear();
        for (const FzGCand &c : cur) {
            FzGOut o;
            fz_generic_step(c, t[index], index, m, pat, max_subs, max_ins, max_dels, max_l, o);
            for (uint32_t i = 0; i < o.nsucc; ++i) nxt.push_back(o.succ[i]);
            for (uint32_t i = 0; i < o.nmatch; ++i) emit(o.mstart[i], o.mend[i], o.mdist[i]);
        }
        cur.swap(nxt);
    }
    for (const FzGCand &c : cur) {
        uint32_t d;
        if (fz_generic_final(c, m, max_dels, max_l, d)) emit(c.start, n, d);
    }
    return cnt;
}

// The same automaton through the packed step the kernel runs (fz_generic_step_packed): candidates as two words,
// outputs in fixed slots.  Also checks, candidate by candidate, that both forms of the step agree (-> -1 if not).
int64_t emul_generic_lp_packed(const uint8_t *p, uint32_t m, const uint8_t *t, uint32_t n, uint32_t max_subs,
                               uint32_t max_ins, uint32_t max_dels, uint32_t max_l, OutRec *out, int64_t cap) {
    std::vector<uint64_t> cur, nxt;
    int64_t cnt = 0;
    bool agree = true;
    auto pat = [&](uint32_t i) -> uint8_t { return p[i]; };
    auto emit = [&](uint32_t se, uint32_t d) {
        if (cnt < cap) { out[cnt].start = se & 0xffffu; out[cnt].end = se >> 16; out[cnt].dist = (int32_t)d; out[cnt].block = -1; }
        ++cnt;
    };
    for (uint32_t index = 0; index < n; ++index) {
        cur.push_back((uint64_t)index);                                // fresh candidate: start = index, everything else 0
        nxt.clear();
        for (uint64_t cw : cur) {
            const uint32_t w0 = (uint32_t)cw, w1 = (uint32_t)(cw >> 32);
            FzGStep st;
            fz_generic_step_packed(w0, w1, t[index], index, m, pat, max_subs, max_ins, max_dels, max_l, st);
            FzGOut o;
            fz_generic_step(fz_gcand_of(w0, w1), t[index], index, m, pat, max_subs, max_ins, max_dels, max_l, o);
            FzGStep ref;
            fz_gstep_from_out(o, ref);
            if (m <= 64u && max_l <= 32u) {                            // the bit-parallel form of the same step (fz_generic_step_bits)
                uint64_t peq = 0;
                for (uint32_t i = 0; i < m; ++i) peq |= (uint64_t)(p[i] == t[index]) << i;
                FzGStep sb;
                fz_generic_step_bits(w0, w1, peq, index, m, max_subs, max_ins, max_dels, max_l, sb);
                agree &= sb.fa == st.fa && sb.fb == st.fb && sb.fc == st.fc && sb.f1 == st.f1 && sb.f2 == st.f2;
                if (st.fa) agree &= sb.a0 == st.a0 && sb.a1 == st.a1;
                if (st.fb) agree &= sb.b0 == st.b0 && sb.b1 == st.b1;
                if (st.fc) agree &= sb.c0 == st.c0 && sb.c1 == st.c1;
                if (st.f1) agree &= sb.m1 == st.m1 && sb.d1 == st.d1;
                if (st.f2) agree &= sb.m2 == st.m2 && sb.d2 == st.d2;
            }
            agree &= st.f1 + st.f2 == ref.f1 + ref.f2;
            if (st.f1 && st.f2) agree &= st.m1 == ref.m1 && st.d1 == ref.d1 && st.m2 == ref.m2 && st.d2 == ref.d2;
            else if (st.f1) agree &= st.m1 == ref.m1 && st.d1 == ref.d1;
            else if (st.f2) agree &= st.m2 == ref.m1 && st.d2 == ref.d1;
            // the struct form fills its slots densely, the packed form by kind: compare the sequences
            uint64_t got[3], want[3];
            uint32_t ng = 0, nw = 0;
            if (st.fa) got[ng++] = st.a0 | ((uint64_t)st.a1 << 32);
            if (st.fb) got[ng++] = st.b0 | ((uint64_t)st.b1 << 32);
            if (st.fc) got[ng++] = st.c0 | ((uint64_t)st.c1 << 32);
            if (ref.fa) want[nw++] = ref.a0 | ((uint64_t)ref.a1 << 32);
            if (ref.fb) want[nw++] = ref.b0 | ((uint64_t)ref.b1 << 32);
            if (ref.fc) want[nw++] = ref.c0 | ((uint64_t)ref.c1 << 32);
            agree &= ng == nw;
            for (uint32_t i = 0; i < ng && i < nw; ++i) agree &= got[i] == want[i];
            for (uint32_t i = 0; i < ng; ++i) nxt.push_back(got[i]);
            if (st.f1) emit(st.m1, st.d1);
            if (st.f2) emit(st.m2, st.d2);
        }
        cur.swap(nxt);
    }
    for (uint64_t cw : cur) {
        uint32_t d;
        const FzGCand c = fz_gcand_of((uint32_t)cw, (uint32_t)(cw >> 32));
        if (fz_generic_final(c, m, max_dels, max_l, d)) emit(c.start | (n << 16), d);
    }
    return agree ? cnt : -1;
}

// The generic n-gram search as the GPU runs it when the rows are ordered on the device (fz_lp_kernel<0> +
// fz_gen_order_kernel + fz_gen_scatter_kernel): n-gram hits arrive in an arbitrary order (here: scrambled with
// `scramble`), every hit runs the packed automaton on its window (window and flush exactly as the kernel
// computes them: fz_segment / fz_hit_in_range_s) and leaves records (hit slot, emission number, se, dist) plus a
// row count; a hit's first row is the sum of the counts of all hits with a smaller key; every record lands at
// first row + emission number as fz_gen_row makes it.  -> rows, or -1 if packed and struct steps ever disagreed.
int64_t emul_generic_ngrams_ordered(const uint8_t *p, uint32_t m, const uint8_t *t, uint64_t n, uint32_t max_subs,
                                    uint32_t max_ins, uint32_t max_dels, uint32_t max_l, uint32_t scramble, uint32_t mode,
                                    OutRec *out, int64_t cap) {
    // mode: bit 0 = the window table (fz_device.h: FzGenDedup — the scan enters every hit, the smallest block of a window
    // leads, members take the leader's rows); bits 1..2 = waves per hit (0: one wave, fz_lp_kernel; 1: two, 2: four —
    // fz_gen_hit_kernel: starts dealt out to the waves, match buffers merged by rank); bits 8.. = member-list length to
    // model (0: the real one), so that windows with more hits than a slot lists are reached with few blocks; bit 3 = the
    // window walked as fz_gen_hit_kernel<W, true> walks it (round 5): the spawning characters up to the closed-form bound
    // n_spawn, then the rest until nothing is alive, the step through the window's equality words (fz_generic_step_bits)
    // bit 4 = the fused consolidation (fz_generic_ngrams_consolidated): every running hit folds its matches into (hull, best)
    // pairs as fz_gen_hit_kernel does (slices of 64 lanes, ballots as loops), a pair with an empty hull is left once per
    // member of the window, and the pairs go through the second stage of the consolidation; -> the consolidated rows
    const bool dedup = mode & 1u;
    const bool two_runs = (mode & 8u) && m <= 64u && max_l <= 32u;
    const bool folded = (mode & 16u) != 0;
    struct Pair { uint64_t key; uint32_t se, dist, win; };
    std::vector<Pair> pairs;
    const uint32_t W = 1u << ((mode >> 1) & 3u);
    const uint32_t members_cap = (mode >> 8) ? std::min<uint32_t>(mode >> 8, FZ_GEN_DEDUP_MEMBERS) : FZ_GEN_DEDUP_MEMBERS;
    const uint32_t k = max_l, L = m / (k + 1);
    if (L == 0) return -2;
    FzScanArgs a;
    memset(&a, 0, sizeof a);
    a.geom.n = n; a.geom.buf_off = 0; a.geom.buf_len = n; a.geom.own_lo = 0; a.geom.own_hi = n;
    a.mode = FZ_MODE_GENERIC; a.m = m; a.k = k; a.L = L; a.abs_lo = 0; a.abs_hi = ~0ull;
    std::vector<uint64_t> hits;
    uint32_t g = 0;
    for (uint32_t s = 0; s + L <= m; s += L, ++g)
        for (uint64_t idx = 0; idx + L <= n; ++idx)
            if (memcmp(t + idx, p + s, L) == 0) {
                const FzSeg sg = fz_segment(a.geom, idx, 0);
                if (fz_hit_in_range_s(a, s, idx, sg)) hits.push_back(fz_hit_pack(g, idx));   // (the scan lists hits in range only)
            }
    // the scan kernel appends hits with atomics: any order
    uint64_t x = 0x9E3779B97F4A7C15ull * (scramble + 1);
    for (size_t i = hits.size(); i > 1; --i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        std::swap(hits[i - 1], hits[x % i]);
    }
    if (hits.size() > FZ_GEN_ORDER_MAX) return -4;
    // fz_gen_claim, hit by hit in list order (the scan does it with atomics as it lists the hits)
    std::vector<uint8_t> table(dedup ? FZ_GEN_DEDUP_BYTES : 8, 0);
    const FzGenDedup dd(reinterpret_cast<uint64_t>(table.data()));
    if (dedup) {
        for (size_t q = 0; q < hits.size(); ++q) {
            const uint32_t blk = fz_hit_block(hits[q]);
            const unsigned long long wk = fz_hit_index(hits[q]) + k - (unsigned long long)blk * L + 1ull;
            uint32_t slot = (uint32_t)((wk * 0x9E3779B97F4A7C15ull) >> 40) & (FZ_GEN_DEDUP_SLOTS - 1u);
            uint32_t at = FZ_GEN_DEDUP_NONE;
            for (uint32_t probe = 0; probe < 32u; ++probe) {
                if (dd.keys[slot] == 0ull || dd.keys[slot] == wk) { dd.keys[slot] = wk; at = slot; break; }
                slot = (slot + 1u) & (FZ_GEN_DEDUP_SLOTS - 1u);
            }
            if (at != FZ_GEN_DEDUP_NONE) {
                const uint32_t pos = dd.nmem[at]++;
                if (pos < members_cap) dd.mem[at * FZ_GEN_DEDUP_MEMBERS + pos] = (uint32_t)q;
                else at = FZ_GEN_DEDUP_NONE;
            }
            if (at != FZ_GEN_DEDUP_NONE) {
                const unsigned long long mine = ~(((unsigned long long)blk << 32) | (unsigned long long)q);
                if (dd.best[at] < mine) dd.best[at] = mine;
            }
            dd.wslot[q] = at;
        }
    }
    struct Rec { uint32_t slot, seq, se, dist; };
    std::vector<Rec> recs;
    std::vector<uint32_t> count(hits.size(), 0);
    auto pat = [&](uint32_t i) -> uint8_t { return p[i]; };
    for (size_t q = 0; q < hits.size(); ++q) {
        const uint64_t hit = hits[q];
        const uint32_t s = fz_hit_block(hit) * L;
        const uint64_t idx = fz_hit_index(hit);
        const FzSeg sg = fz_segment(a.geom, idx, 0);
        if (!fz_hit_in_range_s(a, s, idx, sg)) continue;
        if (dedup && dd.wslot[q] != FZ_GEN_DEDUP_NONE && dd.leader(dd.wslot[q]) != (uint32_t)q) continue;   // its window's leader runs
        const uint64_t reach = (uint64_t)s + k;
        const uint64_t w0 = idx - sg.sa > reach ? idx - reach : sg.sa;
        uint64_t w1 = idx - s + m + k;
        if (w1 > sg.se) w1 = sg.se;
        const uint32_t wlen = (uint32_t)(w1 - w0);
        // every wave: its share of the starts over the whole window, matches buffered as (step << 48 | dist << 32 | se)
        std::vector<std::vector<uint64_t>> mbuf(W);
        for (uint32_t wave = 0; wave < W; ++wave) {
            std::vector<uint64_t> cur, nxt;
            if (two_runs) {
                const uint32_t dlim = max_dels < max_l ? max_dels : max_l;     // (the kernel's own expressions)
                const uint32_t n_spawn = wlen + dlim >= m ? (wlen + dlim - m + 1u < wlen ? wlen + dlim - m + 1u : wlen) : 0u;
                auto one_char = [&](uint32_t index) {
                    uint64_t peq = 0;
                    for (uint32_t i = 0; i < m; ++i) peq |= (uint64_t)(p[i] == t[w0 + index]) << i;
                    nxt.clear();
                    for (uint64_t cw : cur) {
                        FzGStep st;
                        fz_generic_step_bits((uint32_t)cw, (uint32_t)(cw >> 32), peq, index, m, max_subs, max_ins, max_dels, max_l, st);
                        if (st.fa) nxt.push_back(st.a0 | ((uint64_t)st.a1 << 32));
                        if (st.fb) nxt.push_back(st.b0 | ((uint64_t)st.b1 << 32));
                        if (st.fc) nxt.push_back(st.c0 | ((uint64_t)st.c1 << 32));
                        if (st.f1) mbuf[wave].push_back((uint64_t)st.m1 | ((uint64_t)st.d1 << 32) | ((uint64_t)index << 48));
                        if (st.f2) mbuf[wave].push_back((uint64_t)st.m2 | ((uint64_t)st.d2 << 32) | ((uint64_t)index << 48));
                    }
                    cur.swap(nxt);
                };
                uint32_t index = 0;
                for (; index < n_spawn; ++index) {
                    if ((index & (W - 1u)) == wave) cur.push_back((uint64_t)index);
                    if (!cur.empty()) one_char(index);
                }
                for (; index < wlen && !cur.empty(); ++index) one_char(index);
            } else
            for (uint32_t index = 0; index < wlen; ++index) {
                if ((index & (W - 1u)) == wave && fz_gen_start_useful(index, wlen, m, max_dels, max_l)) cur.push_back((uint64_t)index);
                nxt.clear();
                for (uint64_t cw : cur) {
                    FzGStep st;
                    fz_generic_step_packed((uint32_t)cw, (uint32_t)(cw >> 32), t[w0 + index], index, m, pat, max_subs, max_ins, max_dels,
                                           max_l, st);
                    if (st.fa) nxt.push_back(st.a0 | ((uint64_t)st.a1 << 32));
                    if (st.fb) nxt.push_back(st.b0 | ((uint64_t)st.b1 << 32));
                    if (st.fc) nxt.push_back(st.c0 | ((uint64_t)st.c1 << 32));
                    if (st.f1) mbuf[wave].push_back((uint64_t)st.m1 | ((uint64_t)st.d1 << 32) | ((uint64_t)index << 48));
                    if (st.f2) mbuf[wave].push_back((uint64_t)st.m2 | ((uint64_t)st.d2 << 32) | ((uint64_t)index << 48));
                }
                cur.swap(nxt);
            }
            for (uint64_t cw : cur) {                            // end-of-window flush
                uint32_t d;
                const FzGCand c = fz_gcand_of((uint32_t)cw, (uint32_t)(cw >> 32));
                if (fz_generic_final(c, m, max_dels, max_l, d))
                    mbuf[wave].push_back((uint64_t)((uint32_t)c.start | (wlen << 16)) | ((uint64_t)d << 32) | ((uint64_t)wlen << 48));
            }
        }
        if (folded) {
            // first stage of consolidate_overlapping_matches on the device (common.py:150-159): every match that overlaps the
            // running hull is folded into it; wave 0 takes the W buffers one after the other
            bool f_have = false;
            uint32_t f_lo = 0, f_hi = 0, f_k1 = 0, f_k2 = 0;
            auto emit_pair = [&]() {
                const uint32_t sl = dedup ? dd.wslot[q] : FZ_GEN_DEDUP_NONE;
                uint32_t copies = 1u;
                if (sl != FZ_GEN_DEDUP_NONE && f_lo == f_hi) copies = std::min<uint32_t>(dd.nmem[sl], members_cap);
                for (uint32_t cpy = 0; cpy < copies; ++cpy) {
                    const uint32_t len = 0xffffu - (f_k1 & 0xffffu);
                    pairs.push_back(Pair{copies > 1u ? hits[dd.mem[sl * FZ_GEN_DEDUP_MEMBERS + cpy]] : hit, f_k2 | ((f_k2 + len) << 16),
                                         f_k1 >> 16, f_lo | (f_hi << 16)});
                }
                f_have = false;
            };
            for (uint32_t w = 0; w < W; ++w) {
                const uint32_t nw = (uint32_t)mbuf[w].size();
                for (uint32_t e0 = 0; e0 < nw; e0 += 64u) {
                    uint32_t rs[64], re[64], k1[64];
                    uint64_t pending = 0;
                    for (uint32_t lane = 0; lane < 64u; ++lane) {
                        const bool have_row = e0 + lane < nw;
                        const uint64_t v = have_row ? mbuf[w][e0 + lane] : 0ull;
                        rs[lane] = (uint32_t)v & 0xffffu; re[lane] = ((uint32_t)v >> 16) & 0xffffu;
                        k1[lane] = (((uint32_t)(v >> 32) & 0xffffu) << 16) | (0xffffu - (re[lane] - rs[lane]));
                        if (have_row) pending |= 1ull << lane;
                    }
                    while (pending) {
                        if (!f_have) {
                            const uint32_t l0 = (uint32_t)__builtin_ctzll(pending);
                            f_lo = rs[l0]; f_hi = re[l0]; f_k1 = k1[l0]; f_k2 = f_lo;
                            f_have = true;
                            pending &= pending - 1ull;
                            continue;
                        }
                        uint64_t ov = 0;
                        uint32_t mlo = ~0u, nhi = ~0u, mk1 = ~0u;
                        for (uint32_t lane = 0; lane < 64u; ++lane)
                            if (((pending >> lane) & 1ull) && !(re[lane] <= f_lo || rs[lane] >= f_hi)) {
                                ov |= 1ull << lane;
                                mlo = std::min(mlo, rs[lane]); nhi = std::min(nhi, ~re[lane]); mk1 = std::min(mk1, k1[lane]);
                            }
                        if (!ov) { emit_pair(); continue; }
                        uint32_t mk2 = ~0u;
                        for (uint32_t lane = 0; lane < 64u; ++lane)
                            if (((ov >> lane) & 1ull) && k1[lane] == mk1) mk2 = std::min(mk2, rs[lane]);
                        const uint32_t mhi = ~nhi;
                        f_lo = mlo < f_lo ? mlo : f_lo;
                        f_hi = mhi > f_hi ? mhi : f_hi;
                        if (mk1 < f_k1 || (mk1 == f_k1 && mk2 < f_k2)) { f_k1 = mk1; f_k2 = mk2; }
                        pending &= ~ov;
                    }
                }
            }
            if (f_have) emit_pair();
            continue;
        }
        // merge by rank: own position + the entries of the other waves with a smaller (step, start)
        auto key_of = [](uint64_t v) { return ((uint32_t)(v >> 48) << 16) | ((uint32_t)v & 0xffffu); };
        uint32_t total = 0;
        for (uint32_t wave = 0; wave < W; ++wave) total += (uint32_t)mbuf[wave].size();
        for (uint32_t wave = 0; wave < W; ++wave)
            for (uint32_t e = 0; e < mbuf[wave].size(); ++e) {
                const uint64_t v = mbuf[wave][e];
                uint32_t rank = e;
                for (uint32_t w2 = 0; w2 < W; ++w2) {
                    if (w2 == wave) continue;
                    uint32_t lo = 0, hi = (uint32_t)mbuf[w2].size();
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (key_of(mbuf[w2][mid]) < key_of(v)) lo = mid + 1; else hi = mid; }
                    rank += lo;
                }
                recs.push_back(Rec{(uint32_t)q, rank, (uint32_t)v, (uint32_t)(v >> 32) & 0xffffu});
            }
        count[q] = total;
    }
    if (folded) {
        // second stage (emit_generic_result + consolidate_hulls, restated without its slices): the pairs by (hull start,
        // zero-length first, input order), overlapping hulls merged with the better of their best rows, survivors by
        // (start, end, dist, block)
        struct H { int64_t h0, h1; FzOutRow best; };
        std::vector<H> hs;
        for (const Pair &pr : pairs) {
            const FzOutRow best = fz_gen_row(pr.key, L, k, 0, pr.se, pr.dist), hull = fz_gen_row(pr.key, L, k, 0, pr.win, 0);
            hs.push_back(H{hull.start, hull.end, best});
        }
        std::stable_sort(hs.begin(), hs.end(), [](const H &a, const H &b) {
            if (a.h0 != b.h0) return a.h0 < b.h0;
            return (a.h1 == a.h0) && (b.h1 != b.h0);
        });
        auto better = [](const FzOutRow &x, const FzOutRow &y) {
            const int64_t lx = x.end - x.start, ly = y.end - y.start;
            return x.dist < y.dist || (x.dist == y.dist && (lx > ly || (lx == ly && (x.start < y.start || (x.start == y.start && x.block < y.block)))));
        };
        std::vector<FzOutRow> best;
        int64_t h0 = 0, h1 = 0;
        for (const H &h : hs) {
            if (!best.empty() && !(h.h1 <= h0 || h.h0 >= h1)) {
                h0 = std::min(h0, h.h0); h1 = std::max(h1, h.h1);
                if (better(h.best, best.back())) best.back() = h.best;
            } else {
                h0 = h.h0; h1 = h.h1;
                best.push_back(h.best);
            }
        }
        std::sort(best.begin(), best.end(), [](const FzOutRow &a, const FzOutRow &b) {
            if (a.start != b.start) return a.start < b.start;
            if (a.end != b.end) return a.end < b.end;
            if (a.dist != b.dist) return a.dist < b.dist;
            return a.block < b.block;
        });
        for (size_t i = 0; i < best.size() && (int64_t)i < cap; ++i) {
            out[i].start = best[i].start; out[i].end = best[i].end; out[i].dist = best[i].dist; out[i].block = best[i].block;
        }
        return (int64_t)best.size();
    }
    auto rows_of = [&](size_t j) -> uint32_t {
        if (!dedup) return count[j];
        const uint32_t sl = dd.wslot[j];
        return count[sl == FZ_GEN_DEDUP_NONE ? j : dd.leader(sl)];
    };
    std::vector<uint64_t> first(hits.size(), 0);                 // fz_gen_order_kernel
    uint64_t nrows = 0;
    for (size_t i = 0; i < hits.size(); ++i) {
        nrows += rows_of(i);
        for (size_t j = 0; j < hits.size(); ++j)
            if (hits[j] < hits[i]) first[i] += rows_of(j);
    }
    std::vector<FzOutRow> rows(nrows);                           // fz_gen_scatter_kernel
    std::vector<uint8_t> written(nrows, 0);
    auto put = [&](uint32_t h, const Rec &r) -> bool {
        const uint64_t pos = first[h] + r.seq;
        if (pos >= rows.size() || written[pos]) return false;
        written[pos] = 1;
        rows[pos] = fz_gen_row(hits[h], L, k, 0, r.se, r.dist);
        return true;
    };
    for (const Rec &r : recs) {
        const uint32_t sl = dedup ? dd.wslot[r.slot] : FZ_GEN_DEDUP_NONE;
        if (sl == FZ_GEN_DEDUP_NONE) { if (!put(r.slot, r)) return -3; continue; }
        if (dd.leader(sl) != r.slot) continue;
        const uint32_t nm = std::min<uint32_t>(dd.nmem[sl], members_cap);
        for (uint32_t i = 0; i < nm; ++i)
            if (!put(dd.mem[sl * FZ_GEN_DEDUP_MEMBERS + i], r)) return -3;
    }
    for (size_t i = 0; i < rows.size(); ++i) if (!written[i]) return -5;
    for (size_t i = 0; i < rows.size() && (int64_t)i < cap; ++i) {
        out[i].start = rows[i].start; out[i].end = rows[i].end; out[i].dist = rows[i].dist; out[i].block = rows[i].block;
    }
    return (int64_t)rows.size();
}

// find_near_matches_levenshtein_linear_programming over a whole sequence through the struct form of the step
// (fz_levlp_step, the statement of levenshtein.py:52-148) and through the slot form the kernel stores from
// (fz_levlp_step_slots): both must emit the same list; -> -1 if the two forms ever disagree.
int64_t emul_lev_lp(const uint8_t *p, uint32_t m, const uint8_t *t, uint32_t n, uint32_t k, OutRec *out, int64_t cap) {
    std::vector<uint64_t> cur, nxt;
    int64_t cnt = 0;
    bool agree = true;
    auto pat = [&](uint32_t i) -> uint8_t { return p[i]; };
    auto emit = [&](uint32_t se, uint32_t d) {
        if (cnt < cap) { out[cnt].start = se & 0xffffu; out[cnt].end = se >> 16; out[cnt].dist = (int32_t)d; out[cnt].block = -1; }
        ++cnt;
    };
    for (uint32_t index = 0; index < n; ++index) {
        nxt.clear();
        const uint8_t ch = t[index];
        // levenshtein.py:75-80: a fresh candidate for the first pattern char (within the budget) equal to ch goes FIRST
        uint32_t f = 0xffffffffu;
        const uint32_t lim = k + 1 < m ? k + 1 : m;
        for (uint32_t i = 0; i < lim; ++i) if (p[i] == ch) { f = i; break; }
        if (f != 0xffffffffu) {
            if (f + 1 == m) emit(index | ((index + 1) << 16), f);
            else { FzGCand c; c.start = (uint16_t)index; c.j = (uint16_t)(f + 1); c.l = (uint8_t)f; c.ns = c.ni = c.nd = 0; uint32_t w0, w1; fz_gcand_words(c, w0, w1); nxt.push_back(w0 | ((uint64_t)w1 << 32)); }
        }
        const bool more_seq = index + 1 < n;
        for (uint64_t cw : cur) {
            const FzGCand c = fz_gcand_of((uint32_t)cw, (uint32_t)(cw >> 32));
            FzGStep st;
            fz_levlp_step_slots((uint32_t)cw, (uint32_t)(cw >> 32), ch, index, more_seq, m, pat, k, st);
            FzGOut o;
            fz_levlp_step(c, ch, index, more_seq, m, pat, k, o);
            FzGStep ref;
            fz_gstep_from_out(o, ref);
            uint64_t got[3], want[3];
            uint32_t ng = 0, nw = 0;
            if (st.fa) got[ng++] = st.a0 | ((uint64_t)st.a1 << 32);
            if (st.fb) got[ng++] = st.b0 | ((uint64_t)st.b1 << 32);
            if (st.fc) got[ng++] = st.c0 | ((uint64_t)st.c1 << 32);
            if (ref.fa) want[nw++] = ref.a0 | ((uint64_t)ref.a1 << 32);
            if (ref.fb) want[nw++] = ref.b0 | ((uint64_t)ref.b1 << 32);
            if (ref.fc) want[nw++] = ref.c0 | ((uint64_t)ref.c1 << 32);
            agree &= ng == nw && st.f2 == 0 && ref.f2 == 0 && st.f1 == ref.f1;
            for (uint32_t i = 0; i < ng && i < nw; ++i) agree &= got[i] == want[i];
            if (st.f1) agree &= st.m1 == ref.m1 && st.d1 == ref.d1;
            for (uint32_t i = 0; i < ng; ++i) nxt.push_back(got[i]);
            if (st.f1) emit(st.m1, st.d1);
        }
        cur.swap(nxt);
    }
    for (uint64_t cw : cur) {
        uint32_t d;
        const FzGCand c = fz_gcand_of((uint32_t)cw, (uint32_t)(cw >> 32));
        if (fz_levlp_final(c, m, k, d)) emit((uint32_t)c.start | (n << 16), d);
    }
    return agree ? cnt : -1;
}

int emul_expand(const uint8_t *sub, uint32_t sublen, const uint8_t *win, uint32_t winlen, uint32_t budget,
                uint32_t *dist, uint32_t *consumed) {
    HostScores sc;
    sc.v.assign(2 * budget + 4, 0);
    auto s = [&](uint32_t i) -> uint8_t { return sub[i]; };
    auto w = [&](uint32_t j) -> uint8_t { return win[j]; };
    return fz_expand(sc, s, sublen, w, winlen, budget, *dist, *consumed) ? 1 : 0;
}

// register-band variant: band K (the search's k) may exceed the budget of this call
int emul_expand_band(int K, const uint8_t *sub, uint32_t sublen, const uint8_t *win, uint32_t winlen, uint32_t budget,
                     uint32_t *dist, uint32_t *consumed) {
    HostScores sc;
    sc.v.assign(2 * (K > (int)budget ? K : budget) + 4, 0);
    auto s = [&](uint32_t i) -> uint8_t { return sub[i]; };
    auto w = [&](uint32_t j) -> uint8_t { return win[j]; };
    return fz_expand_any<FZ_REG_BAND_MAX>(sc, (uint32_t)K, s, sublen, w, winlen, budget, *dist, *consumed) ? 1 : 0;
}

// The lane-per-DP-cell expansion of fz_kernels.h (fz_wf_rows + fz_wf_pick: fz_verify_wf_kernel and the fused form in the
// scan kernel), restated with a loop over the GW lanes of a group where the device uses DPP: lane gl holds the band cell
// D[i][i + gl - K] of row i; the upper neighbour is the cell of lane gl + 1 (INF beyond the group), the left-neighbour
// recurrence is a prefix-min over the lanes of (value + GW - gl), the bottom row is reduced to (min, LAST arg-min) from
// the column-0 baseline.  `pick_budget` / `pick_winlen` may be smaller than what the rows ran with (the device never
// does that since the side-by-side variant was dropped, but the property the comment in fz_wf_pick states is checked).
int emul_wf_expand(int GW, uint32_t K, const uint8_t *sub, uint32_t sublen, const uint8_t *win, uint32_t winlen, uint32_t budget,
                   uint32_t pick_winlen, uint32_t pick_budget, uint32_t *dist, uint32_t *consumed) {
    const uint32_t INF = 0x3fffu;
    if (GW != 16 && GW != 32 && GW != 64) return -1;
    if (2 * K + 1 > (uint32_t)GW) return -1;
    std::vector<uint32_t> cell(GW), jv(GW), nv(GW);
    for (int gl = 0; gl < GW; ++gl) {
        const bool act = (uint32_t)gl <= 2 * K;
        jv[gl] = act ? (uint32_t)(gl - (int)K) : 0x7fff0000u;
        cell[gl] = jv[gl] <= winlen ? jv[gl] : INF;
    }
    for (uint32_t i = 1; i <= sublen; ++i) {
        const uint8_t pc = sub[i - 1];
        for (int gl = 0; gl < GW; ++gl) {
            // the character of column j = i + d - 1 ... the device reads one element outside the window for cells that are
            // forced anyway; here those cells take a character that never equals anything
            jv[gl] += 1u;                                         // the lane's column in row i
            const uint32_t j = jv[gl];
            const bool in_win = j >= 1 && j <= winlen;
            const uint32_t chr = in_win ? win[j - 1] : 0x100u;
            const uint32_t up = gl + 1 < GW ? cell[gl + 1] : INF;
            uint32_t v = std::min(cell[gl] + (chr != pc ? 1u : 0u), up + 1u);
            if (j == 0u) v = i;                                   // column 0: D[i][0] = i
            const bool bad = j > winlen;
            nv[gl] = bad ? INF : v;
        }
        // v[gl] = min_{e <= gl} (a[e] + gl - e): prefix-min of a[e] + (GW - e), minus (GW - gl)
        uint32_t run = 0xffffffffu;
        for (int gl = 0; gl < GW; ++gl) {
            run = std::min(run, nv[gl] + (uint32_t)(GW - gl));
            uint32_t v = run - (uint32_t)(GW - gl);
            if (jv[gl] > winlen) v = INF;
            cell[gl] = v;
        }
        if ((i & 3u) == 0u) {                                     // row minima never decrease
            bool any = false;
            for (int gl = 0; gl < GW; ++gl) any = any || cell[gl] <= budget;
            if (!any) break;
        }
    }
    // bottom row -> (best, last arg-min) over the columns 1 .. pick_winlen from the column-0 baseline
    uint32_t key = 0xffffffffu;
    for (int gl = 0; gl < GW; ++gl) {
        const int jb = (int)sublen + gl - (int)K;
        const bool validj = (uint32_t)gl <= 2 * K && jb >= 1 && jb <= (int)pick_winlen;
        if (validj) key = std::min(key, (cell[gl] << 8) | (255u - (uint32_t)gl));
    }
    uint32_t best = sublen, arg = 0;
    if (key != 0xffffffffu && (key >> 8) <= sublen) {
        best = key >> 8;
        arg = (uint32_t)((int)sublen + (int)(255u - (key & 255u)) - (int)K);
    }
    *dist = best;
    *consumed = arg;
    return best <= pick_budget ? 1 : 0;
}

}  // extern "C"

// The bit-vector expansion (fz_device.h: fz_bits_column / fz_expand_bits): one piece on its own, its Peq words built for
// the piece alone (row i at bit 64 NW - sublen + i).
template <int NW>
static int expand_bits_nw(const uint8_t *sub, uint32_t sublen, const uint8_t *win, uint32_t winlen, uint32_t budget,
                          uint32_t *dist, uint32_t *consumed) {
    typedef typename FzBitsWord<NW>::T T;
    if (sublen > FZ_BITS_WIDTH(NW)) return -1;
    std::vector<T> tab(256, (T)0);
    for (uint32_t i = 0; i < sublen; ++i) tab[sub[i]] |= (T)1 << (FZ_BITS_WIDTH(NW) - sublen + i);
    auto peq = [&](uint8_t c) -> T { return tab[c]; };
    auto w = [&](uint32_t j) -> uint8_t { return win[j]; };
    return fz_expand_bits<NW>(peq, sublen, w, winlen, budget, *dist, *consumed) ? 1 : 0;
}

template <int NW>
struct HostPeq {
    typedef typename FzBitsWord<NW>::T T;
    std::vector<T> tab[2];
    HostPeq(const uint8_t *p, uint32_t m) {
        tab[0].assign(256, (T)0); tab[1].assign(256, (T)0);
        for (uint32_t q = 0; q < m; ++q) {
            tab[0][p[q]] |= (T)1 << fz_bits_fwd_bit<NW>(m, q);
            tab[1][p[q]] |= (T)1 << fz_bits_rev_bit<NW>(m, q);
        }
    }
    uint32_t table(uint32_t side) const { return side; }
    T at(uint32_t h, uint32_t ch) const { return tab[h][ch]; }
};

// emul_search's Levenshtein form with the per-hit logic of the fused bit-vector verification (fz_verify_lev_bits): both
// pieces of every hit out of the two whole-pattern tables, selected by their top-bit masks.
template <int NW>
static int64_t search_bits_nw(const uint8_t *p, uint32_t m, const uint8_t *t, uint64_t n, uint32_t k,
                              uint64_t buf_off, uint64_t buf_len, uint64_t own_lo, uint64_t own_hi, OutRec *out, int64_t cap) {
    const uint32_t L = m / (k + 1);
    if (L == 0 || m > FZ_BITS_WIDTH(NW)) return -1;
    std::vector<uint8_t> shard(buf_len + 64, 0xEE);
    memcpy(shard.data(), t + buf_off, buf_len);
    const HostPeq<NW> peq(p, m);
    int64_t cnt = 0;
    uint32_t g = 0;
    const int64_t N = (int64_t)n;
    for (uint32_t s = 0; s + L <= m; s += L, ++g) {
        int64_t lo = (int64_t)s - (int64_t)k; if (lo < 0) lo = 0; if (lo > N) lo = N;
        int64_t hi = N - (int64_t)m + s + L + k; if (hi > N) hi = N; if (hi < lo) hi = lo;
        for (int64_t idx = lo; idx + (int64_t)L <= hi; ++idx) {
            if ((uint64_t)idx < own_lo || (uint64_t)idx >= own_hi) continue;
            if (memcmp(t + idx, p + s, L) != 0) continue;
            // the window base the kernel uses: the dword-aligned buffer position at or below max(0, idx - s - k)
            const uint64_t reach = (uint64_t)s + k;
            uint64_t wlo = (uint64_t)idx > reach ? (uint64_t)idx - reach : 0;
            if (wlo < buf_off) wlo = buf_off;
            const uint64_t wbase = buf_off + ((wlo - buf_off) & ~(uint64_t)3);
            // (the loop reads up to two positions past either end of an expansion's window and ignores them: poison here)
            auto txt = [&](uint32_t o) -> uint8_t {
                const uint64_t at = wbase + (uint64_t)(int64_t)(int32_t)o - buf_off;
                return at < shard.size() ? shard[at] : (uint8_t)0xEE;
            };
            FzRec rec;
            if (!fz_verify_lev_bits<NW>(peq, txt, wbase, 0, n, m, k, L, s, (uint64_t)idx, true, rec)) continue;
            if (cnt < cap) {
                out[cnt].start = idx - (int64_t)rec.l;
                out[cnt].end = idx + L + rec.r;
                out[cnt].dist = (int32_t)rec.dist;
                out[cnt].block = (int32_t)g;
            }
            ++cnt;
        }
    }
    return cnt;
}

extern "C" {
int emul_expand_bits(int NW, const uint8_t *sub, uint32_t sublen, const uint8_t *win, uint32_t winlen, uint32_t budget,
                     uint32_t *dist, uint32_t *consumed) {
    return NW == 1 ? expand_bits_nw<1>(sub, sublen, win, winlen, budget, dist, consumed)
         : NW == 4 ? expand_bits_nw<4>(sub, sublen, win, winlen, budget, dist, consumed)
                   : expand_bits_nw<2>(sub, sublen, win, winlen, budget, dist, consumed);
}
int64_t emul_search_bits(int NW, const uint8_t *p, uint32_t m, const uint8_t *t, uint64_t n, uint32_t k,
                         uint64_t buf_off, uint64_t buf_len, uint64_t own_lo, uint64_t own_hi, OutRec *out, int64_t cap) {
    return NW == 1 ? search_bits_nw<1>(p, m, t, n, k, buf_off, buf_len, own_lo, own_hi, out, cap)
         : NW == 4 ? search_bits_nw<4>(p, m, t, n, k, buf_off, buf_len, own_lo, own_hi, out, cap)
                   : search_bits_nw<2>(p, m, t, n, k, buf_off, buf_len, own_lo, own_hi, out, cap);
}
}
