"""-m gpu: every BASELINE.json configuration at its STATED size (1 GiB per GPU), ordered raw streams
bit-exact against the oracle on the full input, plus the 64-bit index path beyond 2^32 bytes.
Workloads: tests/workloads.py; the oracle needs 5-10 s per configuration."""
import numpy as np
import pytest

import oracle
from tests import workloads

pytestmark = pytest.mark.gpu

GIB = 1 << 30


def _rows(arr):
    return [tuple(int(x) for x in r) for r in arr.tolist()]


def test_config1_dna_1gib_levenshtein(engine):
    """configs[1]: 1 GiB random DNA, |p| = 20, max_l_dist = 2 (the headline workload), raw stream and public API."""
    import fuzzysearch_amd as fa
    seq, pat, planted = workloads.cfg2(GIB, 1024)
    p = pat.tobytes()
    h = engine.upload(seq)
    got = _rows(engine.lev_ngrams(h, p, 2, as_array=True))
    st = engine.stats()
    h.release()
    exp = oracle.lev_ngrams_raw(p, seq.tobytes(), 2)
    assert got == exp
    assert len(planted) >= 1000 and len(exp) >= len(planted)
    assert st["bytes_scanned"] == GIB and st["ngram_hits"] > 700000
    res = fa.resident(seq)
    api = fa.find_near_matches(p, res, max_l_dist=2)
    res.release()
    assert [(m.start, m.end, m.dist) for m in api] == oracle.consolidate(exp)
    assert all(bytes(m.matched) == seq[m.start:m.end].tobytes() for m in api[:50])


def test_config2_ascii_1gib_substitutions(engine):
    """configs[2]: 1 GiB over 65 ASCII symbols, |p| = 32, <= 3 substitutions (bytes input)."""
    seq, pat, planted = workloads.cfg3(GIB, 1024)
    p = pat.tobytes()
    h = engine.upload(seq)
    got = _rows(engine.subs_ngrams(h, p, 3, as_array=True))
    h.release()
    exp = oracle.subs_ngrams_raw(p, seq.tobytes(), 3)
    assert got == exp
    assert len(exp) >= len(planted) >= 1000


def test_config3_utf8_1gib_wide_band_and_generic(engine):
    """configs[3]: 1 GiB UTF-8 text as bytes, |p| = 64: (a) max_l_dist = 5 -> Levenshtein n-grams with the
    lane-per-cell verification, (b) limits (5, 2, 2, 5) -> generic search (SURVEY.md trap 4)."""
    seq, pat, planted = workloads.cfg4(GIB, 1024)
    p, t = pat.tobytes(), seq.tobytes()
    h = engine.upload(seq)
    got_a = _rows(engine.lev_ngrams(h, p, 5, as_array=True))
    got_b = _rows(engine.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True))
    h.release()
    exp_a = oracle.lev_ngrams_raw(p, t, 5)
    assert got_a == exp_a
    assert len(exp_a) >= len(planted) >= 1000
    exp_b = oracle.generic_ngrams_raw(p, t, 5, 2, 2, 5)
    assert got_b == exp_b
    assert len(exp_b) > 100000


@pytest.mark.parametrize("k", [2, 3])
def test_beyond_4gib_indices(engine, k):
    """64-bit index paths: 4.5 GiB = nine copies of one 512 MiB block, variants planted beyond 2^32 and
    across 2^32.  Size-independent property: the matches inside copy i are those of copy 0 shifted by
    i * 512 MiB (the oracle only has to run on one block and on the planted tail).  k = 2: the register band (the
    headline instance); k = 3: the fused bit-vector form (round 6)."""
    block = 512 << 20
    copies = 9
    n = block * copies
    pattern = workloads.dna(20, 1)
    p = pattern.tobytes()
    base = workloads.dna(block, 900)
    seq = np.tile(base, copies)
    tail0 = 8 * block + (block >> 1)                     # 4.25 GiB: the second half of the last copy
    assert tail0 > (1 << 32)
    planted = workloads.plant_variants(seq[tail0:], pattern, 256, 17)
    seq[(1 << 32) - 10:(1 << 32) + 10] = pattern         # a match straddling 2^32 (= start of copy 8)
    h = engine.upload(seq)
    got = _rows(engine.lev_ngrams(h, p, k, as_array=True))
    h.release()
    keys = [(g, s) for (s, e, d, g) in got]
    assert keys == sorted(keys)                          # block-major, ascending: reference order
    margin = 64
    exp0 = [r for r in oracle.lev_ngrams_raw(p, base.tobytes(), k) if margin <= r[0] and r[1] <= block - margin]
    assert len(exp0) > 0
    for i in range(8):                                   # copies 0..7 are untouched away from their seams
        lo, hi = i * block, (i + 1) * block
        inside = sorted((s - lo, e - lo, d, g) for (s, e, d, g) in got if lo + margin <= s and e <= hi - margin)
        assert inside == sorted(exp0), i
    exp_tail = oracle.lev_ngrams_raw(p, seq[tail0:].tobytes(), k)
    got_tail = [(s - tail0, e - tail0, d, g) for (s, e, d, g) in got if s >= tail0 + margin]
    assert got_tail == [r for r in exp_tail if r[0] >= margin]
    assert len(planted) >= 200 and len(got_tail) >= len(planted)
    assert any(s == (1 << 32) - 10 and e == (1 << 32) + 10 and d == 0 for (s, e, d, g) in got)


def test_has_near_match_leaves_a_4gib_scan_early(engine):
    """has_near_match_* (substitutions_only.py:218-233, generic_search.py:240-253 return at the first match): on a long
    sequence the flag-only searches scan in growing pieces (64 MiB, then 4 x as much each time) and stop behind the first
    piece with a match — a match in the first MiB of a 4 GiB sequence answers in a few percent of a full scan, for the
    substitutions form and (scan of the piece + automaton on its hits) for the generic one.
    Same flag, no match: the whole buffer is scanned and the answer is False."""
    import time
    n = 4 << 30
    seq = np.empty(n, dtype=np.uint8)
    for i in range(4):
        seq[i << 30:(i + 1) << 30] = workloads.dna(1 << 30, 700 + i)
    pattern = workloads.text65(24, 5)                   # letters a DNA sequence does not have: no match unless planted
    p = pattern.tobytes()
    h = engine.upload(seq)
    del seq

    def timed(fn, reps):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        return (time.perf_counter() - t0) / reps, r
    t_full, none = timed(lambda: engine.subs_ngrams_any(h, p, 2), 10)
    assert none is False
    h.release()
    seq2 = np.empty(n, dtype=np.uint8)
    for i in range(4):
        seq2[i << 30:(i + 1) << 30] = workloads.dna(1 << 30, 700 + i)
    seq2[500000:500000 + len(pattern)] = pattern
    h2 = engine.upload(seq2)
    del seq2
    t_hit, found = timed(lambda: engine.subs_ngrams_any(h2, p, 2), 10)
    t_gen, found_g = timed(lambda: engine.generic_ngrams_any(h2, p, 2, 1, 1, 2), 10)
    h2.release()
    assert found is True and found_g is True
    # round 5: 0.21 against 0.84 ms (25 %: one launch whose late workgroups skipped their tiles, bounded by their finish
    # tickets); the generic form paid a full scan + the automaton launch.  Round 6 (pieces): one 64 MiB search each.
    assert t_hit < 0.08 * t_full, (t_hit, t_full)
    assert t_gen < 0.5 * t_full, (t_gen, t_full)


def test_has_near_match_pieces_find_matches_anywhere(engine):
    """The flag-only searches' pieces (64 MiB, 256 MiB, the rest of a 640 MiB sequence): a match across a piece boundary —
    its n-gram hits in one piece, its window reaching into the other —, a match in the last bytes, and no match."""
    n = 640 << 20
    base = workloads.dna(n, 910)
    pattern = workloads.text65(24, 6)                   # letters a DNA sequence does not have
    p = pattern.tobytes()
    for where in ((64 << 20) - 11, (64 << 20) - 24, (320 << 20) - 7, n - len(pattern), None):
        seq = base.copy()
        if where is not None:
            seq[where:where + len(pattern)] = pattern
            seq[where + 5] = ord("A")                   # one substitution: blocks on both sides of it still hit
        h = engine.upload(seq)
        want = where is not None
        assert engine.subs_ngrams_any(h, p, 2) is want, where
        assert engine.generic_ngrams_any(h, p, 2, 1, 1, 2) is want, where
        if where is not None:                           # the searches proper are untouched by the pieces
            assert [r[0] for r in engine.subs_ngrams(h, p, 2)][:1] == [where]
        h.release()
