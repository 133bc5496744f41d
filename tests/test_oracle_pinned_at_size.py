"""The oracle — the expected side of every GPU assertion — pinned against the REFERENCE's compiled path where the GPU
is actually judged: 16 MiB of every BASELINE workload at its stated pattern shape (tests/workloads.py, the generators
of the 1 GiB parity tests and of bench.py), raw streams ordered-equal; wide budgets (m up to 400, k up to 40: the
regimes of tests/test_gpu_limits.py); and the consolidation of the large generic stream against the reference's
consolidate_overlapping_matches, tie-aware.  (tests/test_oracle_vs_reference.py covers thousands of tiny cases: n <= 60.)
Build container only: /root/reference does not travel to the GPU box."""
import random

import numpy as np
import pytest

import oracle
from fuzzysearch_amd import _native
from oracle import ref_loader
from tests import golden_io, workloads

pytestmark = pytest.mark.skipif(not ref_loader.have_reference_package(),
                                reason="/root/reference or oracle/_ref not available (GPU box)")

N = 16 << 20


@pytest.fixture(scope="module")
def ref():
    return ref_loader.load_reference_package()


def _triples(ms):
    return [(x.start, x.end, x.dist) for x in ms]


def test_configs1_dna_levenshtein_m20_k2(ref):
    from fuzzysearch.levenshtein_ngram import find_near_matches_levenshtein_ngrams as ref_fn
    seq, pat, planted = workloads.cfg2(N, 64)
    p, t = pat.tobytes(), seq.tobytes()
    exp = _triples(ref_fn(p, t, 2))
    got = oracle.lev_ngrams_raw(p, t, 2)
    assert [r[:3] for r in got] == exp and len(exp) >= len(planted) > 40
    # the CPU leg of bench.py (oracle/ref_glue: the reference's natives under a restated outer loop) is the same stream
    from oracle import ref_glue
    assert ref_glue.lev_ngrams_raw(p, t, 2) == exp
    # ... and the public API's answer, tie-aware
    api = _triples(ref.find_near_matches(p, t, max_l_dist=2))
    assert golden_io.equal_modulo_ties(oracle.consolidate(got), api, got)


def test_configs2_ascii_substitutions_m32_k3(ref):
    from fuzzysearch.substitutions_only import _subs_only_fnm_ngram_byteslike as ref_fn
    seq, pat, planted = workloads.cfg3(N, 64)
    p, t = pat.tobytes(), seq.tobytes()
    exp = list(ref_fn(p, t, 3))
    got = oracle.subs_ngrams_raw(p, t, 3)
    assert [r[0] for r in got] == exp and len(exp) >= len(planted) > 40
    # the reference's own Match construction + group-list-order reduction (substitutions_only.py:258-282)
    api = _triples(ref.find_near_matches(p, t, max_substitutions=3, max_insertions=0, max_deletions=0))
    best, _hull = oracle.group_best(got)
    assert [(s, e, d) for (s, e, d, _g) in best] == api
    mine = _native.group_best([tuple(r) for r in got])
    assert [tuple(r[:3]) for r in mine] == api


def test_configs3a_utf8_levenshtein_m64_k5(ref):
    from fuzzysearch.levenshtein_ngram import find_near_matches_levenshtein_ngrams as ref_fn
    seq, pat, planted = workloads.cfg4(N, 64)
    p, t = pat.tobytes(), seq.tobytes()
    exp = _triples(ref_fn(p, t, 5))
    got = oracle.lev_ngrams_raw(p, t, 5)
    assert [r[:3] for r in got] == exp and len(exp) > 100
    assert {d for (_s, _e, d) in exp} >= {0, 1, 2, 3, 4, 5}             # every distance of the 0..5-edit plants occurs


def test_configs3b_utf8_generic_m64_limits_and_its_consolidation(ref):
    from fuzzysearch.common import LevenshteinSearchParams, consolidate_overlapping_matches
    from fuzzysearch.generic_search import find_near_matches_generic_ngrams as ref_fn
    seq, pat, _planted = workloads.cfg4(N, 64)
    p, t = pat.tobytes(), seq.tobytes()
    sp = LevenshteinSearchParams(5, 2, 2, 5)
    ref_raw = list(ref_fn(p, t, sp))
    exp = _triples(ref_raw)
    got = oracle.generic_ngrams_raw(p, t, 5, 2, 2, 5)
    assert [r[:3] for r in got] == exp and len(exp) > 3000                # ordered list, duplicates included
    # consolidation of the large stream: the reference's own function on its own Match objects, against the oracle's and
    # the product's host-side consolidation (fz_consolidate needs no device), tie-aware
    ref_cons = _triples(consolidate_overlapping_matches(ref_raw))
    mine = oracle.consolidate(got)
    assert len(mine) == len(ref_cons) > 30
    assert golden_io.equal_modulo_ties(mine, ref_cons, got)
    prod = [tuple(r[:3]) for r in _native.consolidate([tuple(r) for r in got])]
    assert golden_io.equal_modulo_ties(prod, ref_cons, got) and prod == mine
    # ... and the reference's public API takes this route for these limits
    api = _triples(ref.find_near_matches(p, t, max_substitutions=5, max_insertions=2, max_deletions=2, max_l_dist=5))
    assert golden_io.equal_modulo_ties(mine, api, got)


def _plant(rnd, t, p, alpha, k):
    v = bytearray(p)
    for _ in range(rnd.randint(0, k)):
        q = rnd.randrange(len(v))
        op = rnd.random()
        if op < 0.4:
            v[q] = rnd.choice(alpha)
        elif op < 0.7 and len(v) > k + 1:
            del v[q]
        else:
            v.insert(q, rnd.choice(alpha))
    at = rnd.randint(0, max(0, len(t) - len(v)))
    t[at:at + len(v)] = v


def test_wide_budgets_long_patterns(ref):
    """m in 60..400, k in 8..40 (Levenshtein n-grams: lane-per-cell and big verification regimes on the GPU), plus the
    generic and substitutions routes at such shapes."""
    from fuzzysearch.common import LevenshteinSearchParams
    from fuzzysearch.generic_search import find_near_matches_generic_ngrams as ref_gen
    from fuzzysearch.levenshtein_ngram import find_near_matches_levenshtein_ngrams as ref_lev
    from fuzzysearch.substitutions_only import _subs_only_fnm_ngram_byteslike as ref_subs
    rnd = random.Random(404)
    n_lev = n_hits = 0
    for it in range(90):
        sigma = rnd.choice([2, 3, 4, 4, 20])
        alpha = bytes(rnd.sample(range(33, 127), sigma))
        m = rnd.randint(60, 400)
        k = rnd.randint(8, 40)
        if m // (k + 1) < 3:                                     # the n-gram regime of the reference's dispatcher (levenshtein.py:19-30)
            continue
        p = bytes(rnd.choice(alpha) for _ in range(m))
        t = bytearray(rnd.choice(alpha) for _ in range(rnd.randint(2000, 6000) if sigma >= 4 else rnd.randint(1200, 2500)))
        for _ in range(rnd.randint(1, 4)):
            _plant(rnd, t, p, alpha, k)
        t = bytes(t)
        exp = _triples(ref_lev(p, t, k))
        assert [r[:3] for r in oracle.lev_ngrams_raw(p, t, k)] == exp, (it, m, k)
        n_lev += 1
        n_hits += bool(exp)
        if it % 3 == 0:
            ks = min(k, 12)
            assert [r[0] for r in oracle.subs_ngrams_raw(p, t, ks)] == list(ref_subs(p, t, ks)), (it, m, ks)
        if it % 4 == 0:
            kg = min(k, 6)
            ms, mi, md = rnd.randint(0, kg), rnd.randint(0, kg), rnd.randint(0, kg)
            sp = LevenshteinSearchParams(ms, mi, md, kg)
            a = sp.unpacked
            if m // (a[3] + 1) >= 1:
                assert [r[:3] for r in oracle.generic_ngrams_raw(p, t[:3000], *a)] == _triples(ref_gen(p, t[:3000], sp)), (it, m, a)
    assert n_lev >= 50 and n_hits >= 40
