"""The synthetic workloads of SURVEY.md §8(d) plant what they say (checked with the oracle at a small size):
cfg 3 variants carry i % 4 substitutions, cfg 4 variants up to 5 mixed edits, and every one is a match."""
import collections

import oracle
from tests import workloads

N = 4 << 20


def test_cfg3_plants_are_matches_with_their_substitution_count():
    seq, pat, planted = workloads.cfg3(N, 256)
    assert len(planted) >= 200
    raw = oracle.subs_ngrams_raw(pat.tobytes(), seq.tobytes(), 3)
    dist_at = {s: d for (s, e, d, g) in raw}
    assert all(dist_at.get(p0) == ne for (p0, ne) in planted)
    assert collections.Counter(ne for (_p, ne) in planted).keys() == {0, 1, 2, 3}


def test_cfg4_plants_cover_zero_to_five_mixed_edits():
    seq, pat, planted = workloads.cfg4(N, 256)
    assert len(planted) >= 200
    assert set(ne for (_p, ne) in planted) == {0, 1, 2, 3, 4, 5}
    best = oracle.consolidate(oracle.lev_ngrams_raw(pat.tobytes(), seq.tobytes(), 5))
    m = len(pat)
    for (p0, ne) in planted:
        near = [d for (s, e, d) in best if abs(s - p0) <= 5]
        assert near and min(near) <= ne, (p0, ne, near)
    # the deep states are exercised: some planted match really needs 4 or 5 edits
    deep = [min(d for (s, e, d) in best if abs(s - p0) <= 5) for (p0, ne) in planted if ne >= 4]
    assert max(deep) >= 4
    generic = oracle.generic_ngrams_raw(pat.tobytes(), seq.tobytes(), 5, 2, 2, 5)
    assert len(generic) > 0 and max(d for (s, e, d, g) in generic) >= 3
