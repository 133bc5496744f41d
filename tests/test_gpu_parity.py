"""-m gpu: the HIP path (through the C-ABI) against the oracle, bit-exact and ordered."""
import random

import numpy as np
import pytest

import oracle
from tests import workloads

pytestmark = pytest.mark.gpu


def _reload_switches():
    from fuzzysearch_amd import _native
    _native.load_library().fz_debug_reload_switches()


def _rand_case(rnd, max_n=300, max_m=24, max_k=4):
    sigma = rnd.choice([2, 2, 3, 4, 4, 20])
    alpha = bytes(rnd.sample(range(33, 127), sigma))
    n = rnd.randint(0, max_n)
    t = bytes(rnd.choice(alpha) for _ in range(n))
    k = rnd.randint(1, max_k)
    m = rnd.randint(k + 1, max_m)
    if rnd.random() < 0.6 and n >= m:
        st = rnd.randint(0, n - m)
        p = bytearray(t[st:st + m])
        for _ in range(rnd.randint(0, k)):
            q = rnd.randrange(len(p))
            op = rnd.random()
            if op < 0.4:
                p[q] = rnd.choice(alpha)
            elif op < 0.7 and len(p) > k + 1:
                del p[q]
            else:
                p.insert(q, rnd.choice(alpha))
        p = bytes(p)
    else:
        p = bytes(rnd.choice(alpha) for _ in range(m))
    return p, t, k


def test_lev_ngrams_raw_random(engine):
    rnd = random.Random(11)
    n_cases = 0
    for _ in range(1500):
        p, t, k = _rand_case(rnd)
        if len(p) // (k + 1) == 0:
            continue
        seq = engine.upload(t)
        got = engine.lev_ngrams(seq, p, k)
        seq.release()
        assert got == oracle.lev_ngrams_raw(p, t, k), (p, t, k)
        n_cases += 1
    assert n_cases > 1000


def test_lev_ngrams_wide_budgets_random(engine):
    """Budgets 5 .. 31 run the lane-per-cell wavefront kernel (16 / 32 / 64 lanes per candidate), larger
    ones the lane-per-candidate LDS ring: every group width, ragged sequence ends, dense repeats."""
    rnd = random.Random(17)
    n_cases = {16: 0, 32: 0, 64: 0, 0: 0}
    for it in range(900):
        sigma = rnd.choice([2, 3, 4, 4, 20])
        alpha = bytes(rnd.sample(range(33, 127), sigma))
        k = rnd.choice([5, 5, 6, 7, 7, 8, 9, 12, 15, 16, 20, 31, 32, 40])
        m = rnd.randint(k + 1, min(4 * (k + 1) + 20, 140))
        n = rnd.randint(0, 500)
        t = bytearray(rnd.choice(alpha) for _ in range(n))
        p = bytes(rnd.choice(alpha) for _ in range(m))
        if n > m + 10 and rnd.random() < 0.7:          # plant an edited copy
            v = bytearray(p)
            for _ in range(rnd.randint(0, k)):
                q = rnd.randrange(len(v))
                op = rnd.random()
                if op < 0.4:
                    v[q] = rnd.choice(alpha)
                elif op < 0.7 and len(v) > 2:
                    del v[q]
                else:
                    v.insert(q, rnd.choice(alpha))
            st = rnd.choice([0, 1, n - len(v) - 1, n - len(v), rnd.randint(0, max(0, n - len(v)))])
            st = max(0, min(st, n - len(v)))
            t[st:st + len(v)] = v
        t = bytes(t)
        seq = engine.upload(t)
        got = engine.lev_ngrams(seq, p, k)
        seq.release()
        assert got == oracle.lev_ngrams_raw(p, t, k), (p, t, k)
        n_cases[16 if k <= 7 else 32 if k <= 15 else 64 if k <= 31 else 0] += 1
    assert all(v > 50 for v in n_cases.values()), n_cases


def test_lev_ngrams_budgets_5_to_8_long_ngrams(engine):
    """Budgets 5 .. 8 with n-grams of 8 bytes or more (the shape of BASELINE configs[3]: long pattern, wide band,
    few n-gram hits) against the oracle: planted edited copies, ragged ends, small alphabets.  (Round 2 measured
    verifying these inside the scan kernel with a register band up to 8: 0.283 instead of 0.206 + 0.031 ms at
    m = 64, k = 5 on 1 GiB of text — a lone candidate's 64-row DP at the end of a wave's life costs more than
    the lane-per-cell kernel needs for all of them; not kept.)"""
    rnd = random.Random(23)
    cases = []
    for it in range(260):
        alpha = bytes(rnd.sample(range(33, 127), rnd.choice([2, 3, 4, 20])))
        k = rnd.choice([5, 6, 7, 8])
        m = rnd.randint(8 * (k + 1), 8 * (k + 1) + 40)
        n = rnd.randint(0, 1500)
        t = bytearray(rnd.choice(alpha) for _ in range(n))
        p = bytes(rnd.choice(alpha) for _ in range(m))
        for _rep in range(3):
            if n > m + 10 and rnd.random() < 0.8:
                v = bytearray(p)
                for _ in range(rnd.randint(0, k)):
                    q = rnd.randrange(len(v))
                    op = rnd.random()
                    if op < 0.4:
                        v[q] = rnd.choice(alpha)
                    elif op < 0.7 and len(v) > 2:
                        del v[q]
                    else:
                        v.insert(q, rnd.choice(alpha))
                st = rnd.choice([0, 1, n - len(v) - 1, n - len(v), rnd.randint(0, max(0, n - len(v)))])
                st = max(0, min(st, n - len(v)))
                t[st:st + len(v)] = v
        cases.append((p, bytes(t), k))
    hits = 0
    for p, t, k in cases:
        seq = engine.upload(t)
        got = engine.lev_ngrams(seq, p, k)
        seq.release()
        assert got == oracle.lev_ngrams_raw(p, t, k), (p, t, k)
        hits += len(got)
    assert hits > 200


def test_subs_ngrams_raw_random(engine):
    rnd = random.Random(12)
    for _ in range(1000):
        p, t, k = _rand_case(rnd)
        if len(p) // (k + 1) == 0:
            continue
        seq = engine.upload(t)
        got = engine.subs_ngrams(seq, p, k)
        seq.release()
        assert got == oracle.subs_ngrams_raw(p, t, k), (p, t, k)


def test_search_exact_random(engine):
    rnd = random.Random(13)
    for _ in range(1000):
        sigma = rnd.choice([1, 2, 3, 4])
        alpha = bytes(rnd.sample(range(65, 91), sigma))
        n = rnd.randint(0, 400)
        t = bytes(rnd.choice(alpha) for _ in range(n))
        m = rnd.randint(1, 14)
        p = bytes(rnd.choice(alpha) for _ in range(m))
        lo = rnd.randint(-3, n + 3)
        hi = rnd.randint(-3, n + 3)
        seq = engine.upload(t)
        got = engine.search_exact(seq, p, max(0, lo), max(0, hi))
        got_all = engine.search_exact(seq, p)
        seq.release()
        assert got == oracle.search_exact(p, t, lo, hi), (p, t, lo, hi)
        assert got_all == oracle.search_exact(p, t), (p, t)


@pytest.mark.parametrize("m,k", [(20, 2), (9, 2), (12, 1), (32, 3), (23, 5), (64, 5), (10, 0 + 1), (40, 12), (64, 8), (100, 20)])
def test_lev_ngrams_medium_dna(engine, m, k):
    """4 MiB of DNA with planted variants: every n-gram length / block count regime."""
    n = 4 << 20
    seq = workloads.dna(n, 99 + m)
    pattern = workloads.dna(m, 5 + k)
    workloads.plant_variants(seq, pattern, 256, 3)
    t, p = seq.tobytes(), pattern.tobytes()
    h = engine.upload(seq)
    got = engine.lev_ngrams(h, p, k)
    h.release()
    exp = oracle.lev_ngrams_raw(p, t, k)
    assert got == exp
    assert len(exp) >= 100


@pytest.mark.parametrize("m,k", [(32, 3), (20, 2), (12, 3), (9, 2)])
def test_subs_ngrams_medium(engine, m, k):
    n = 4 << 20
    seq = workloads.dna(n, 199 + m)
    pattern = workloads.dna(m, 15 + k)
    workloads.plant_variants(seq, pattern, 256, 4)
    t, p = seq.tobytes(), pattern.tobytes()
    h = engine.upload(seq)
    got = engine.subs_ngrams(h, p, k)
    h.release()
    assert got == oracle.subs_ngrams_raw(p, t, k)


def test_dense_and_degenerate_inputs(engine):
    """Pathological hit densities: queue overflow -> slow tile path, record/hit buffer growth."""
    t = b'A' * 200000
    h = engine.upload(t)
    for p, k in [(b'A' * 9, 2), (b'AAAAAAAAAAAB', 2), (b'AAAB', 0 + 1 - 1 + 1)]:
        if len(p) // (k + 1) == 0:
            continue
        got = engine.lev_ngrams(h, p, k)
        assert got == oracle.lev_ngrams_raw(p, t, k)
        got = engine.subs_ngrams(h, p, k)
        assert got == oracle.subs_ngrams_raw(p, t, k)
    assert engine.search_exact(h, b'AAA') == oracle.search_exact(b'AAA', t)
    h.release()
    # zero bytes in pattern and sequence (the device buffer is zero padded)
    t = bytes(1000) + b'\x01\x00\x00\x02' + bytes(50)
    h = engine.upload(t)
    for p, k in [(bytes(12), 2), (b'\x00\x00\x01\x00\x00\x02\x00\x00\x00', 2)]:
        assert engine.lev_ngrams(h, p, k) == oracle.lev_ngrams_raw(p, t, k)
        assert engine.subs_ngrams(h, p, k) == oracle.subs_ngrams_raw(p, t, k)
    assert engine.search_exact(h, bytes(5)) == oracle.search_exact(bytes(5), t)
    h.release()
    # empty and shorter-than-pattern sequences
    for t in (b'', b'ACG', b'ACGTACGTAC'):
        h = engine.upload(t)
        assert engine.lev_ngrams(h, b'ACGTACGTACGT', 2) == oracle.lev_ngrams_raw(b'ACGTACGTACGT', t, 2)
        assert engine.subs_ngrams(h, b'ACGTACGTACGT', 2) == oracle.subs_ngrams_raw(b'ACGTACGTACGT', t, 2)
        assert engine.search_exact(h, b'ACG') == oracle.search_exact(b'ACG', t)
        h.release()


def test_result_staging_mode_switches(engine):
    """Small result sets are written by the kernels straight into pinned host memory, large ones go
    through the device buffer + copy; alternate between the two and compare every call."""
    dense = b'ACGT' * 30000                       # ~90 000 raw matches: beyond the host staging buffer
    sparse = workloads.dna(1 << 18, 5).tobytes()
    p, k = b'ACGTACGTACGTAC', 2
    hd, hs = engine.upload(dense), engine.upload(sparse)
    want_d, want_s = oracle.lev_ngrams_raw(p, dense, k), oracle.lev_ngrams_raw(p, sparse, k)
    assert len(want_d) > 20000
    for _ in range(3):
        assert engine.lev_ngrams(hs, p, k) == want_s
        assert engine.lev_ngrams(hd, p, k) == want_d
        assert engine.subs_ngrams(hs, p, k) == oracle.subs_ngrams_raw(p, sparse, k)
    assert engine.lev_ngrams(hs, p, k) == want_s
    hd.release(); hs.release()


def test_blocks_split_over_several_launches(engine, monkeypatch):
    """Blocks whose hashes cannot share the slot table go to separate scan launches; FZ_MAX_BLOCKS
    forces that split (1 and 2 blocks per launch) for the fused, the emit + verify and the
    substitutions paths; only the last launch publishes the counters."""
    t = workloads.dna(1 << 20, 21).tobytes()
    pats = [(workloads.dna(20, 1).tobytes(), 2), (workloads.dna(23, 5).tobytes(), 5), (workloads.dna(32, 3).tobytes(), 3)]
    for p, _k in pats:
        t = t[:5000] + p + t[5000:70000] + p[:7] + p[8:] + t[70000:]
    h = engine.upload(t)
    for cap in ("1", "2"):
        monkeypatch.setenv("FZ_MAX_BLOCKS", cap)
        _reload_switches()
        for p, k in pats:
            assert engine.lev_ngrams(h, p, k) == oracle.lev_ngrams_raw(p, t, k)
            assert engine.subs_ngrams(h, p, k) == oracle.subs_ngrams_raw(p, t, k)
        assert engine.stats()["filter_launches"] >= 2
    monkeypatch.delenv("FZ_MAX_BLOCKS")
    _reload_switches()
    h.release()


def test_begin_end_split_equals_the_synchronous_call(engine):
    """fz_lev_ngrams_begin / _end: same stream as fz_lev_ngrams, also across a result-buffer overflow; up to two
    searches in flight, delivered oldest first; other searches are refused meanwhile."""
    t = workloads.dna(1 << 20, 31).tobytes()
    p, p2 = t[5000:5020], t[70000:70024]
    dense = b'ACGT' * 30000
    h, hd = engine.upload(t), engine.upload(dense)
    want, want2 = oracle.lev_ngrams_raw(p, t, 2), oracle.lev_ngrams_raw(p2, t, 3)
    for _ in range(3):
        engine.lev_ngrams_begin(h, p, 2)
        with pytest.raises(ValueError):
            engine.lev_ngrams(h, p, 2)
        engine.lev_ngrams_begin(h, p2, 3)                      # second search in flight: the other result slot
        with pytest.raises(ValueError):
            engine.lev_ngrams_begin(h, p, 2)                   # a third one is refused
        assert engine.lev_ngrams_end() == want                 # oldest first
        engine.lev_ngrams_begin(h, p, 2)                       # ... and its slot is free again
        assert engine.lev_ngrams_end() == want2
        assert engine.lev_ngrams_end() == want
    with pytest.raises(ValueError):
        engine.lev_ngrams_end()
    # a steady two-deep pipeline, as bench.py drives it
    engine.lev_ngrams_begin(h, p, 2)
    for i in range(6):
        engine.lev_ngrams_begin(h, p2 if i % 2 == 0 else p, 3 if i % 2 == 0 else 2)
        assert engine.lev_ngrams_end() == (want if i % 2 == 0 else want2)
    assert engine.lev_ngrams_end() == want
    # result-buffer overflows (the dense search leaves direct mode, grows the device buffers and re-runs) with a
    # second search in flight, before and behind it
    pd = b'ACGTACGTACGTAC'
    want_d = oracle.lev_ngrams_raw(pd, dense, 2)
    assert len(want_d) > 20000
    engine.lev_ngrams_begin(hd, bytearray(pd), 2)              # the pattern buffer is copied by _begin
    engine.lev_ngrams_begin(h, p, 2)
    assert engine.lev_ngrams_end() == want_d
    engine.lev_ngrams_begin(hd, pd, 2)                         # launched (or deferred) while not in direct mode
    assert engine.lev_ngrams_end() == want
    engine.lev_ngrams_begin(h, p2, 3)
    assert engine.lev_ngrams_end() == want_d
    assert engine.lev_ngrams_end() == want2
    assert engine.lev_ngrams(h, p, 2) == want
    h.release(); hd.release()


def test_automaton_candidate_lists_in_hbm(engine, monkeypatch):
    """Candidate sets that outgrow LDS move to per-workgroup lists in HBM (slow, pathological inputs
    only).  FZ_CAND_LDS_MAX forces that path at ordinary sizes; afterwards the context goes back to
    LDS lists."""
    rnd = random.Random(5)
    t = bytes(rnd.choice(b'ab') for _ in range(600))
    p = bytes(rnd.choice(b'ab') for _ in range(30))
    t_ok = workloads.dna(1 << 16, 9).tobytes()
    p_ok = t_ok[300:340]
    h, h_ok = engine.upload(t), engine.upload(t_ok)
    want_ok = oracle.generic_ngrams_raw(p_ok, t_ok, 2, 1, 1, 2)
    monkeypatch.setenv("FZ_CAND_LDS_MAX", "16")
    _reload_switches()
    assert engine.generic_ngrams(h, p, 3, 3, 3, 3) == oracle.generic_ngrams_raw(p, t, 3, 3, 3, 3)
    assert engine.generic_ngrams(h_ok, p_ok, 2, 1, 1, 2) == want_ok
    assert engine.lev_lp(h, p[:8], 3) == oracle.lev_lp_raw(p[:8], t, 3)
    assert engine.generic_lp(h, p[:8], 2, 2, 2, 3) == oracle.generic_lp_raw(p[:8], t, 2, 2, 2, 3)
    monkeypatch.delenv("FZ_CAND_LDS_MAX")
    _reload_switches()
    assert engine.generic_ngrams(h_ok, p_ok, 2, 1, 1, 2) == want_ok
    assert engine.generic_ngrams(h, p, 3, 3, 3, 3) == oracle.generic_ngrams_raw(p, t, 3, 3, 3, 3)
    h.release(); h_ok.release()


def test_sharded_equals_unsharded(engine):
    """Two shards with (m + k) halos, hits owned by index == one sequence."""
    rnd = random.Random(21)
    n = 1 << 20
    seq = workloads.dna(n, 77)
    pattern = workloads.dna(20, 1)
    workloads.plant_variants(seq, pattern, 512, 9)
    p, k = pattern.tobytes(), 2
    t = seq.tobytes()
    exp = oracle.lev_ngrams_raw(p, t, k)
    halo = len(p) + k
    for _ in range(6):
        cut = rnd.choice([0, 1, halo - 1, halo, n // 2, n - halo, n - 1, n, rnd.randint(0, n)])
        a = engine.upload_shard(seq[:min(n, cut + halo)], 0, 0, cut, n)
        lo = max(0, cut - halo)
        b = engine.upload_shard(seq[lo:], lo, cut, n, n)
        got = engine.lev_ngrams(a, p, k) + engine.lev_ngrams(b, p, k)
        a.release()
        b.release()
        got.sort(key=lambda r: r[3])          # stable: block major, rank order keeps idx ascending
        assert got == exp, cut


def test_public_api_matches_oracle(engine):
    import fuzzysearch_amd as fa
    rnd = random.Random(31)
    for _ in range(300):
        p, t, k = _rand_case(rnd, max_n=200)
        m = len(p)
        if m // (k + 1) < 3:
            continue
        got = fa.find_near_matches(p, t, max_l_dist=k)
        exp = oracle.consolidate(oracle.lev_ngrams_raw(p, t, k))
        assert [(x.start, x.end, x.dist) for x in got] == exp, (p, t, k)
        assert all(x.matched == t[x.start:x.end] for x in got)
    assert fa.find_near_matches(b'PATTERN', b'---PATERN---', max_l_dist=1) == \
        [fa.Match(3, 9, 1, b'PATERN')]
    assert fa.find_near_matches('PATTERN', '---PATERN---', max_l_dist=1) == [fa.Match(3, 9, 1, 'PATERN')]


def test_fused_search_and_reduction_entry_points(engine):
    """Round 4: fz_lev_ngrams_consolidated / fz_subs_ngrams_best (search + the strategy class's reduction in one C-ABI
    call) against fz_consolidate / fz_group_best of the raw streams and against the oracle; find_near_matches takes
    them and builds the Match objects in C straight from the result buffer — for bytes, bytearray, str and resident
    sequences, `matched` has the type the sequence's own slicing gives."""
    import fuzzysearch_amd as fa
    rnd = random.Random(47)
    n_lev = n_subs = 0
    for _ in range(400):
        p, t, k = _rand_case(rnd, max_n=400)
        if len(p) // (k + 1) < 1:
            continue
        h = engine.upload(t)
        raw = oracle.lev_ngrams_raw(p, t, k)
        assert [r[:3] for r in engine.lev_ngrams_consolidated(h, p, k)] == oracle.consolidate(raw), (p, t, k)
        rs = oracle.subs_ngrams_raw(p, t, k)
        best, _hull = oracle.group_best(rs)
        assert [r[:3] for r in engine.subs_ngrams_best(h, p, k)] == [b[:3] for b in best], (p, t, k)
        h.release()
        n_lev += bool(raw)
        n_subs += bool(rs)
        if len(p) // (k + 1) >= 3:
            got = fa.find_near_matches(p, t, max_substitutions=k, max_insertions=0, max_deletions=0)
            assert [(x.start, x.end, x.dist) for x in got] == [b[:3] for b in best], (p, t, k)
            assert all(type(x.matched) is bytes and x.matched == t[x.start:x.end] for x in got)
    assert n_lev > 150 and n_subs > 100
    # sequence types: the slices are what sequence[start:end] gives
    t = workloads.dna(1 << 16, 8).tobytes()
    p = t[5000:5020]
    want = oracle.consolidate(oracle.lev_ngrams_raw(p, t, 2))
    for seq, typ in ((t, bytes), (bytearray(t), bytearray), (t.decode('latin-1'), str), (np.frombuffer(t, dtype=np.uint8), np.ndarray),
                     (memoryview(t), memoryview)):
        pat = p.decode('latin-1') if typ is str else p
        got = fa.find_near_matches(pat, seq, max_l_dist=2)
        assert [(x.start, x.end, x.dist) for x in got] == want
        assert all(type(x.matched) is typ and bytes(x.matched if typ is not str else x.matched.encode('latin-1')) == t[x.start:x.end] for x in got)
    res = fa.resident(t)
    got = fa.find_near_matches(p, res, max_l_dist=2)
    assert [(x.start, x.end, x.dist) for x in got] == want and all(x.matched == t[x.start:x.end] for x in got)
    got = fa.find_near_matches(p, res, max_substitutions=2, max_insertions=0, max_deletions=0)
    best, _hull = oracle.group_best(oracle.subs_ngrams_raw(p, t, 2))
    assert [(x.start, x.end, x.dist) for x in got] == [b[:3] for b in best]
    res.release()
    # str input on the substitutions route: every window sorted by start (the reference's pure-Python form), not fused
    ts = t.decode('latin-1')
    got = fa.find_near_matches(p.decode('latin-1'), ts, max_substitutions=2, max_insertions=0, max_deletions=0)
    assert [x.start for x in got] == sorted({r[0] for r in oracle.subs_ngrams_raw(p, t, 2)})


def test_two_streams_pipeline_equals_the_synchronous_calls():
    """fz_set_streams(2): the younger of two fused searches in flight scans on a stream and a counter block of its own.
    Same streams as the synchronous calls for Levenshtein / substitutions searches in any mix, different patterns and
    sequences, across a result-buffer overflow (the search leaves direct mode: back on the one stream) and back."""
    from fuzzysearch_amd import _native
    eng = _native.Engine([0])
    try:
        eng.set_streams(2)
        with pytest.raises(ValueError):
            eng.set_streams(3)
        t = workloads.dna(8 << 20, 33).tobytes()
        t2 = workloads.text65(4 << 20, 34).tobytes()
        pats = [(t[5000:5020], 2), (t[70000:70024], 3), (t[123456:123470], 1), (t[999:1031], 4)]
        for q, (p, k) in enumerate(pats):
            t = t[:200000 * (q + 1)] + p + t[200000 * (q + 1) + len(p):]
        h, h2 = eng.upload(t), eng.upload(t2)
        want = {(p, k): (oracle.lev_ngrams_raw(p, t, k), oracle.subs_ngrams_raw(p, t, k)) for p, k in pats}
        p2 = t2[777:809]
        want2 = oracle.subs_ngrams_raw(p2, t2, 3)
        rnd = random.Random(5)
        inflight = []
        for step in range(120):
            p, k = rnd.choice(pats)
            kind = rnd.choice(["lev", "subs", "other"])
            if kind == "lev":
                eng.lev_ngrams_begin(h, p, k); inflight.append(want[(p, k)][0])
            elif kind == "subs":
                eng.subs_ngrams_begin(h, p, k); inflight.append(want[(p, k)][1])
            else:
                eng.subs_ngrams_begin(h2, p2, 3); inflight.append(want2)
            if len(inflight) == 2:
                assert eng.search_end() == inflight.pop(0), step
        while inflight:
            assert eng.search_end() == inflight.pop(0)
        # more records than the pinned staging buffer holds: the search falls back to the device buffer (one stream), a
        # small one afterwards returns to direct mode and to the second stream
        dense = b"ACGT" * 30000
        pd_, hd = b"ACGTACGTACGTAC", eng.upload(b"ACGT" * 30000)
        wd = oracle.lev_ngrams_raw(pd_, dense, 2)
        assert len(wd) > 20000
        p, k = pats[0]
        for _ in range(3):
            eng.lev_ngrams_begin(hd, pd_, 2)
            eng.lev_ngrams_begin(h, p, k)
            assert eng.search_end() == wd and eng.search_end() == want[(p, k)][0]
            eng.lev_ngrams_begin(h, p, k)
            eng.lev_ngrams_begin(h, p, k)
            assert eng.search_end() == want[(p, k)][0] and eng.search_end() == want[(p, k)][0]
        eng.set_streams(1)
        assert eng.lev_ngrams(h, p, k) == want[(p, k)][0]
    finally:
        eng.close()


def test_generic_ngrams_raw_random(engine):
    """a11: the candidate-set automaton, ordered raw stream == oracle (App. A.3)."""
    rnd = random.Random(41)
    n_cases = 0
    for _ in range(1200):
        sigma = rnd.choice([2, 3, 4, 4, 20])
        alpha = bytes(rnd.sample(range(33, 127), sigma))
        t = bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, 200)))
        m = rnd.randint(2, 20)
        if rnd.random() < 0.6 and len(t) >= m:
            st = rnd.randint(0, len(t) - m)
            p = bytearray(t[st:st + m])
            for _ in range(rnd.randint(0, 3)):
                q = rnd.randrange(len(p))
                op = rnd.random()
                if op < 0.4:
                    p[q] = rnd.choice(alpha)
                elif op < 0.7 and len(p) > 2:
                    del p[q]
                else:
                    p.insert(q, rnd.choice(alpha))
            p = bytes(p)
        else:
            p = bytes(rnd.choice(alpha) for _ in range(m))
        ms, mi, md = rnd.randint(0, 3), rnd.randint(0, 3), rnd.randint(0, 3)
        ml = rnd.randint(1, max(1, ms + mi + md))
        ms, mi, md = min(ms, ml), min(mi, ml), min(md, ml)
        if len(p) // (ml + 1) == 0:
            continue
        seq = engine.upload(t)
        got = engine.generic_ngrams(seq, p, ms, mi, md, ml)
        seq.release()
        assert got == oracle.generic_ngrams_raw(p, t, ms, mi, md, ml), (p, t, (ms, mi, md, ml))
        n_cases += 1
    assert n_cases > 800


@pytest.mark.parametrize("m,limits", [(20, (2, 1, 1, 2)), (20, (3, 1, 2, 3)), (40, (5, 2, 2, 5)), (64, (5, 2, 2, 5)), (12, (1, 1, 0, 1))])
def test_generic_ngrams_medium(engine, m, limits):
    n = 2 << 20
    seq = workloads.dna(n, 300 + m) if m < 64 else workloads.text65(n, 300 + m)
    pattern = workloads.dna(m, 30 + m) if m < 64 else workloads.text65(m, 30 + m)
    workloads.plant_variants(seq, pattern, 128, 13, workloads.DNA if m < 64 else workloads.TEXT65)
    t, p = seq.tobytes(), pattern.tobytes()
    h = engine.upload(seq)
    got = engine.generic_ngrams(h, p, *limits)
    h.release()
    exp = oracle.generic_ngrams_raw(p, t, *limits)
    assert got == exp
    assert len(exp) > 50


def test_generic_more_hits_than_the_device_orders(engine):
    """3.7e4 n-gram hits (4-letter text, 4-byte n-grams): above FZ_GEN_ORDER_MAX = 16384 the rows are ordered by the
    host from records whose `win` field holds the hit slot; below it (the other generic tests) on the device."""
    seq = workloads.dna(3 << 20, 5)
    t = seq.tobytes()
    p = t[1000:1012]
    h = engine.upload(seq)
    for limits in ((1, 1, 1, 2), (2, 0, 1, 2)):
        got = engine.generic_ngrams(h, p, *limits)
        assert engine.stats()["ngram_hits"] > 16384
        want = oracle.generic_ngrams_raw(p, t, *limits)
        assert got == want, limits
        cons = engine.generic_ngrams_consolidated(h, p, *limits)
        assert [r[:3] for r in cons] == oracle.consolidate(want), limits
    h.release()
    # consolidated with more (hull, best) pairs than the pinned staging buffer holds (16 384): 2e4 copies of the
    # pattern, three hits each — the search that overflows the buffer is run again through the device buffer, the next
    # one stays there, and a search with few pairs returns to direct mode
    rnd = random.Random(77)
    p = bytes(rnd.choice(b"ACGT") for _ in range(12))
    t = b"".join(p + bytes(rnd.choice(b"ACGT") for _ in range(8)) for _ in range(20000))
    h = engine.upload(t)
    want = oracle.consolidate(oracle.generic_ngrams_raw(p, t, 1, 1, 1, 2))
    for _ in range(2):
        cons = engine.generic_ngrams_consolidated(h, p, 1, 1, 1, 2)
        assert engine.stats()["raw_matches"] > 16384
        assert [r[:3] for r in cons] == want
    q = bytes(rnd.choice(b"ACGT") for _ in range(24))
    for _ in range(2):
        small = engine.generic_ngrams_consolidated(h, q, 2, 1, 1, 3)
        assert [r[:3] for r in small] == oracle.consolidate(oracle.generic_ngrams_raw(q, t, 2, 1, 1, 3))
    h.release()


def test_generic_public_api(engine):
    import fuzzysearch_amd as fa
    rnd = random.Random(43)
    n = 0
    for _ in range(200):
        alpha = bytes(rnd.sample(range(65, 91), rnd.choice([2, 3, 4])))
        t = bytes(rnd.choice(alpha) for _ in range(rnd.randint(20, 150)))
        m = rnd.randint(9, 16)
        st = rnd.randint(0, len(t) - m)
        p = bytearray(t[st:st + m])
        p[rnd.randrange(m)] = rnd.choice(alpha)
        p = bytes(p)
        got = fa.find_near_matches(p, t, max_substitutions=2, max_insertions=1, max_deletions=1, max_l_dist=2)
        raw = oracle.generic_ngrams_raw(p, t, 2, 1, 1, 2)
        assert [(x.start, x.end, x.dist) for x in got] == oracle.consolidate(raw), (p, t)
        n += 1
    assert n == 200


@pytest.mark.parametrize("dev_threads", [True, False])
def test_multi_device_context_on_one_gpu(dev_threads, monkeypatch):
    """fz_create([0, 0, 0]): three device states on the same GPU exercise the single-process
    multi-device path (shard + halo + concurrent launches + host merge) end to end — with one host worker thread per
    device (round 4: every worker enqueues, collects and orders its own shard, the caller merges the ordered rows)
    and with the calling thread doing everything (FZ_NO_DEV_THREADS=1)."""
    from fuzzysearch_amd import _native
    if not dev_threads:
        monkeypatch.setenv("FZ_NO_DEV_THREADS", "1")
    _reload_switches()                          # (the library reads its switches once; see fz_debug_reload_switches)
    eng = _native.Engine([0, 0, 0])
    monkeypatch.delenv("FZ_NO_DEV_THREADS", raising=False)
    _reload_switches()
    rnd = random.Random(51)
    for n in (0, 5, 100, 5000, 1 << 20):
        seq = workloads.dna(n, 500 + n % 97)
        pattern = workloads.dna(20, 1)
        if n >= 4096:
            workloads.plant_variants(seq, pattern, 64, 3)
            for cut in (n // 3, 2 * n // 3, n // 3 + 1):      # variants straddling the shard cuts
                seq[cut - 9:cut + 11] = pattern
        t, p = seq.tobytes(), pattern.tobytes()
        h = eng.upload(seq)
        assert eng.lev_ngrams(h, p, 2) == oracle.lev_ngrams_raw(p, t, 2), n
        assert eng.subs_ngrams(h, p, 2) == oracle.subs_ngrams_raw(p, t, 2), n
        assert eng.search_exact(h, p[:7]) == oracle.search_exact(p[:7], t), n
        gen = oracle.generic_ngrams_raw(p, t, 2, 1, 1, 2)
        assert eng.generic_ngrams(h, p, 2, 1, 1, 2) == gen, n
        # round 3: the consolidated / flag-only / pipelined forms and the wide-band verification over shards
        assert [r[:3] for r in eng.generic_ngrams_consolidated(h, p, 2, 1, 1, 2)] == oracle.consolidate(gen), n
        assert eng.generic_ngrams_any(h, p, 2, 1, 1, 2) == (len(gen) > 0), n
        assert eng.subs_ngrams_any(h, p, 2) == (len(oracle.subs_ngrams_raw(p, t, 2)) > 0), n
        eng.generic_ngrams_begin(h, p, 2, 1, 1, 2)
        eng.generic_ngrams_begin(h, p, 2, 1, 1, 2)
        assert eng.search_end() == gen and eng.search_end() == gen, n
        eng.subs_ngrams_begin(h, p, 2)
        eng.lev_ngrams_begin(h, p, 2)
        assert eng.search_end() == oracle.subs_ngrams_raw(p, t, 2) and eng.search_end() == oracle.lev_ngrams_raw(p, t, 2), n
        if n >= 4096:
            p6 = (p + p[:5])[:25]
            assert eng.lev_ngrams(h, p6, 6) == oracle.lev_ngrams_raw(p6, t, 6), n
            # many records per shard (beyond the pinned staging buffer on the dense one): the copy mode per worker
            dense = np.frombuffer(b"ACGT" * (n // 4), dtype=np.uint8)
            hd = eng.upload(dense)
            pd_ = b"ACGTACGTACGTAC"
            assert eng.lev_ngrams(hd, pd_, 2) == oracle.lev_ngrams_raw(pd_, dense.tobytes(), 2)
            assert eng.lev_ngrams(h, p, 2) == oracle.lev_ngrams_raw(p, t, 2), n
            hd.release()
        h.release()
    assert eng.stats()["n_devices"] == 3
    eng.close()


def test_lp_fallbacks_raw_random(engine):
    """(f)3: the reference's linear-programming fallbacks (short patterns), ordered lists == oracle."""
    rnd = random.Random(61)
    for _ in range(1500):
        sigma = rnd.choice([2, 3, 4, 4, 20])
        alpha = bytes(rnd.sample(range(33, 127), sigma))
        t = bytes(rnd.choice(alpha) for _ in range(rnd.choice([0, 1, 5, 40, 300, 700])))
        p = bytes(rnd.choice(alpha) for _ in range(rnd.randint(1, 10)))
        k = rnd.randint(0, 4)
        seq = engine.upload(t)
        try:
            assert engine.lev_lp(seq, p, k) == oracle.lev_lp_raw(p, t, k), (p, t, k)
            assert engine.subs_lp(seq, p, k) == oracle.subs_lp_raw(p, t, k), (p, t, k)
            ms, mi, md = rnd.randint(0, 3), rnd.randint(0, 3), rnd.randint(0, 3)
            ml = rnd.randint(0, ms + mi + md)
            ms, mi, md = min(ms, ml), min(mi, ml), min(md, ml)
            assert engine.generic_lp(seq, p, ms, mi, md, ml) == oracle.generic_lp_raw(p, t, ms, mi, md, ml), (p, t, (ms, mi, md, ml))
        finally:
            seq.release()


def test_lp_fallbacks_medium_and_api(engine):
    import fuzzysearch_amd as fa
    n = 1 << 18
    seq = workloads.dna(n, 71)
    t = seq.tobytes()
    h = engine.upload(seq)
    for p, k in [(b'ACGTA', 1), (b'GATTACA', 2), (b'ACG', 1)]:
        assert engine.lev_lp(h, p, k) == oracle.lev_lp_raw(p, t, k)
        assert engine.subs_lp(h, p, k) == oracle.subs_lp_raw(p, t, k)
    assert engine.generic_lp(h, b'GATTACA', 2, 1, 1, 2) == oracle.generic_lp_raw(b'GATTACA', t, 2, 1, 1, 2)
    h.release()
    small = t[:5000]
    got = fa.find_near_matches(b'GATTACA', small, max_l_dist=2)              # 7 // 3 < 3 -> LP route
    assert [(x.start, x.end, x.dist) for x in got] == oracle.consolidate(oracle.lev_lp_raw(b'GATTACA', small, 2))
    got = fa.find_near_matches(b'GATTA', small, max_substitutions=1, max_insertions=0, max_deletions=0)
    assert [(x.start, x.end, x.dist) for x in got] == [r[:3] for r in oracle.subs_lp_raw(b'GATTA', small, 1)]
    got = fa.find_near_matches(b'GATTACA', small, max_substitutions=2, max_insertions=1, max_deletions=1, max_l_dist=2)
    assert [(x.start, x.end, x.dist) for x in got] == oracle.consolidate(oracle.generic_lp_raw(b'GATTACA', small, 2, 1, 1, 2))
    assert len(fa.find_near_matches(b'ab', b'xxxx', max_l_dist=2)) == 5     # k >= m: levenshtein.py:62-65


def test_threads_share_the_default_engine(engine):
    """ctypes drops the GIL during fz_* calls; the per-engine lock must keep a shared fz_ctx sane."""
    import threading
    import fuzzysearch_amd as fa
    seq = workloads.dna(1 << 18, 77)
    pats = [workloads.dna(20, 100 + i) for i in range(8)]
    for p in pats:
        workloads.plant_variants(seq, p, 16, int(p[0]) + 1)
    t = seq.tobytes()
    exp = [oracle.consolidate(oracle.lev_ngrams_raw(p.tobytes(), t, 2)) for p in pats]
    res, errs = [None] * len(pats), []

    def work(i):
        try:
            for _ in range(5):
                res[i] = [(m.start, m.end, m.dist) for m in fa.find_near_matches(pats[i].tobytes(), t, max_l_dist=2)]
        except Exception as e:          # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(pats))]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    assert res == exp


def test_has_near_match_flag_only_paths(engine):
    """has_near_match_* (substitutions_only.py:139-145, :218-233; generic_search.py:240-253) through the flag-only
    entry points: same answer as "the raw stream is non-empty", for matches at the start, the end, nowhere."""
    import fuzzysearch_amd.generic_search as gs
    import fuzzysearch_amd.substitutions_only as so
    from fuzzysearch_amd.common import LevenshteinSearchParams
    rnd = random.Random(41)
    seen = {True: 0, False: 0}
    for it in range(300):
        p, t, k = _rand_case(rnd, max_n=400)
        h = engine.upload(t)
        if len(p) // (k + 1):
            want = len(oracle.subs_ngrams_raw(p, t, k)) > 0
            assert engine.subs_ngrams_any(h, p, k) == want, (p, t, k)
            lim = (rnd.randint(0, k), rnd.randint(0, k), rnd.randint(0, k), k)
            assert engine.generic_ngrams_any(h, p, *lim) == (len(oracle.generic_ngrams_raw(p, t, *lim)) > 0), (p, t, lim)
            seen[want] += 1
        assert engine.subs_lp_any(h, p, k) == (len(oracle.subs_lp_raw(p, t, k)) > 0), (p, t, k)
        h.release()
    assert seen[True] > 50 and seen[False] > 50
    # large input, match only in the last tile / nowhere; the public functions
    seq = workloads.text65(64 << 20, 5)
    pat = workloads.text65(32, 6)
    t_no = seq.tobytes()
    assert so.has_near_match_substitutions_ngrams(pat.tobytes(), t_no, 3) is False
    assert gs.has_near_match_generic_ngrams(pat.tobytes(), t_no, LevenshteinSearchParams(2, 1, 1, 3)) is False
    v = pat.copy()
    v[5] = ord('#'); v[20] = ord('#')
    seq[-40:-8] = v
    t_yes = seq.tobytes()
    assert so.has_near_match_substitutions_ngrams(pat.tobytes(), t_yes, 3) is True
    assert so.has_near_match_substitutions(pat.tobytes(), t_yes, 3) is True
    assert so.has_near_match_substitutions_lp(pat.tobytes()[:6], t_yes, 3) is True
    assert gs.has_near_match_generic_ngrams(pat.tobytes(), t_yes, LevenshteinSearchParams(2, 1, 1, 3)) is True
    seq[100:132] = pat                                            # ... and at the very start: later workgroups skip their tiles
    h = engine.upload(seq)
    assert engine.subs_ngrams_any(h, pat.tobytes(), 3) is True
    st_any = engine.stats()
    full = engine.subs_ngrams(h, pat.tobytes(), 3)
    h.release()
    assert len(full) >= 2 and st_any["bytes_scanned"] == len(seq)


def test_exact_routes_return_lazy_streams(engine):
    """ExactSearch and the k == 0 routes hand RawMatches (an index array) to consolidation: same results as before."""
    import fuzzysearch_amd as fa
    from fuzzysearch_amd.common import RawMatches, LevenshteinSearchParams
    t = (b"abcabcabd" * 5000) + b"xyz"
    p = b"abcabd"
    exp = oracle.search_exact(p, t)
    res = fa.ExactSearch.search(p, t, LevenshteinSearchParams(0, 0, 0, 0))
    assert isinstance(res, RawMatches) and len(res) == len(exp) == 5000
    assert [(m.start, m.end, m.dist, bytes(m.matched)) for m in res[:3]] == [(i, i + 6, 0, p) for i in exp[:3]]
    api = fa.find_near_matches(p, t, max_l_dist=0)
    assert [(m.start, m.end, m.dist) for m in api] == [(i, i + 6, 0) for i in exp]
    assert fa.find_near_matches(p, t, max_substitutions=0, max_insertions=0, max_deletions=0) == api
    from fuzzysearch_amd.search_exact import search_exact
    assert search_exact(p, t) == exp and search_exact(p, t, 10, 100) == [i for i in exp if 10 <= i and i + 6 <= 100]


def test_generic_consolidated_equals_consolidation_of_the_raw_stream(engine):
    """fz_generic_ngrams_consolidated (first stage of consolidate_overlapping_matches on the device: every hit's
    matches folded into hull / best pairs) == fz_consolidate(fz_generic_ngrams) row for row, and the public API takes it."""
    import fuzzysearch_amd as fa
    from fuzzysearch_amd import _native
    rnd = random.Random(61)
    n_rows = 0
    for it in range(400):
        p, t, k = _rand_case(rnd, max_n=600, max_m=40, max_k=5)
        if len(p) // (k + 1) == 0:
            continue
        lim = (rnd.randint(0, k), rnd.randint(0, k), rnd.randint(0, k), k)
        h = engine.upload(t)
        raw = engine.generic_ngrams(h, p, *lim, as_array=True)
        want = _native.consolidate_array(raw).tolist()
        got = engine.generic_ngrams_consolidated(h, p, *lim, as_array=True).tolist()
        h.release()
        assert got == want, (p, t, lim)                              # block included: equal rows of different hits -> smallest block
        n_rows += len(got)
    assert n_rows > 100
    # zero-length matches (deletions only) and a medium text with planted variants
    seq, pat, _pl = workloads.cfg4(8 << 20, 64)
    h = engine.upload(seq)
    for lim in ((5, 2, 2, 5), (1, 0, 3, 3), (0, 2, 0, 2), (2, 2, 2, 6)):
        raw = engine.generic_ngrams(h, pat.tobytes(), *lim, as_array=True)
        want = _native.consolidate_array(raw).tolist()
        got = engine.generic_ngrams_consolidated(h, pat.tobytes(), *lim, as_array=True).tolist()
        assert got == want, lim
    h.release()
    res = fa.find_near_matches(pat.tobytes(), seq.tobytes(), max_substitutions=5, max_insertions=2, max_deletions=2, max_l_dist=5)
    raw = oracle.generic_ngrams_raw(pat.tobytes(), seq.tobytes(), 5, 2, 2, 5)
    assert [(m.start, m.end, m.dist) for m in res] == oracle.consolidate(raw)
    assert all(bytes(m.matched) == seq[m.start:m.end].tobytes() for m in res)


def test_two_searches_in_flight_for_every_kind(engine):
    """fz_subs_ngrams_begin / fz_generic_ngrams_begin + fz_search_end: the two-deep pipeline delivers, oldest first,
    exactly what the synchronous calls return; generic searches run on two lanes (second stream, second buffers)."""
    seq, pat, _pl = workloads.cfg4(16 << 20, 64)
    p = pat.tobytes()
    p2 = workloads.utf8_text(48, 9).tobytes()
    h = engine.upload(seq)
    want_g = engine.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True)
    want_g2 = engine.generic_ngrams(h, p2, 3, 1, 1, 3, as_array=True)
    want_c = engine.generic_ngrams_consolidated(h, p, 5, 2, 2, 5, as_array=True)
    want_s = engine.subs_ngrams(h, p, 5, as_array=True)
    want_l = engine.lev_ngrams(h, p, 3, as_array=True)
    assert len(want_g) > 1000 and len(want_c) > 10
    for rep in range(3):
        engine.generic_ngrams_begin(h, p, 5, 2, 2, 5)
        engine.generic_ngrams_begin(h, p2, 3, 1, 1, 3)
        assert np.array_equal(engine.search_end(as_array=True), want_g)
        engine.generic_ngrams_begin(h, p, 5, 2, 2, 5, consolidated=True)
        assert np.array_equal(engine.search_end(as_array=True), want_g2)
        engine.generic_ngrams_begin(h, p, 5, 2, 2, 5)
        assert np.array_equal(engine.search_end(as_array=True), want_c)
        assert np.array_equal(engine.search_end(as_array=True), want_g)
    # substitutions-only searches share the Levenshtein pipeline and may be mixed with it
    engine.subs_ngrams_begin(h, p, 5)
    engine.lev_ngrams_begin(h, p, 3)
    assert np.array_equal(engine.search_end(as_array=True), want_s)
    engine.subs_ngrams_begin(h, p, 5)
    assert np.array_equal(engine.search_end(as_array=True), want_l)
    with pytest.raises(ValueError):
        engine.generic_ngrams_begin(h, p, 5, 2, 2, 5)             # not together with the other kinds
    assert np.array_equal(engine.search_end(as_array=True), want_s)
    with pytest.raises(ValueError):
        engine.search_end()                                        # nothing in flight
    # an automaton overflow (candidate lists) in a pipelined search is re-run when it is collected
    t = (b"ab" * 4000)
    pg = b"abababababab"
    h2 = engine.upload(t)
    want = engine.generic_ngrams(h2, pg, 2, 2, 2, 3, as_array=True)
    engine.generic_ngrams_begin(h2, pg, 2, 2, 2, 3)
    engine.generic_ngrams_begin(h2, pg, 2, 2, 2, 3)
    assert np.array_equal(engine.search_end(as_array=True), want)
    assert np.array_equal(engine.search_end(as_array=True), want)
    h2.release()
    h.release()
