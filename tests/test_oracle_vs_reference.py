"""Random differentials: the oracle (oracle/fz_oracle.c) against the REFERENCE's own compiled
C/Cython path (oracle/_ref + /root/reference, build container only).  This is what pins the
oracle beyond the reference's unit tests — in particular the last-arg-min tie rule of `expand`
(SURVEY.md trap 2), which the reference's own KATs do not distinguish."""
import os
import random
import subprocess
import sys

import pytest

import oracle
from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.have_reference_package(),
                                reason="/root/reference or oracle/_ref not available (GPU box)")


@pytest.fixture(scope="module")
def ref():
    if os.environ.get("PYTHONHASHSEED") != "0":
        # raw streams are hash-seed independent; only consolidated ties are not, and those are
        # compared tie-aware.  Nothing to enforce here.
        pass
    return ref_loader.load_reference_package()


def _case(rnd, max_n=60, max_m=14, max_k=3):
    sigma = rnd.choice([2, 2, 3, 4])
    alpha = bytes(rnd.sample(range(65, 91), sigma))
    n = rnd.randint(0, max_n)
    t = bytes(rnd.choice(alpha) for _ in range(n))
    k = rnd.randint(1, max_k)
    m = rnd.randint(k + 1, max_m)
    if rnd.random() < 0.5 and n >= m:
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


def test_lev_ngrams_raw(ref):
    from fuzzysearch.levenshtein_ngram import find_near_matches_levenshtein_ngrams as ref_fn
    rnd = random.Random(1)
    n = 0
    for _ in range(6000):
        p, t, k = _case(rnd)
        if len(p) // (k + 1) == 0:
            continue
        assert [(x.start, x.end, x.dist) for x in ref_fn(p, t, k)] == \
            [r[:3] for r in oracle.lev_ngrams_raw(p, t, k)], (p, t, k)
        n += 1
    assert n > 4000


def test_expand_tie_rule(ref):
    from fuzzysearch.levenshtein_ngram import _expand
    rnd = random.Random(2)
    for _ in range(20000):
        alpha = bytes(rnd.sample(range(65, 91), rnd.choice([2, 3])))
        sub = bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, 16)))
        win = bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, 20)))
        k = rnd.randint(0, 5)
        assert tuple(_expand(sub, win, k)) == oracle.expand(sub, win, k), (sub, win, k)


def test_subs_ngrams_raw(ref):
    from fuzzysearch.substitutions_only import _subs_only_fnm_ngram_byteslike as ref_fn
    rnd = random.Random(3)
    for _ in range(6000):
        p, t, k = _case(rnd, max_k=4)
        assert list(ref_fn(p, t, k)) == [r[0] for r in oracle.subs_ngrams_raw(p, t, k)], (p, t, k)


def test_generic_raw(ref):
    from fuzzysearch.common import LevenshteinSearchParams
    from fuzzysearch.generic_search import (find_near_matches_generic_linear_programming as ref_lp,
                                            find_near_matches_generic_ngrams as ref_ng)
    rnd = random.Random(4)
    for _ in range(2500):
        alpha = bytes(rnd.sample(range(65, 91), rnd.choice([2, 3, 4])))
        t = bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, 40)))
        p = bytes(rnd.choice(alpha) for _ in range(rnd.randint(1, 12)))
        ms, mi, md = rnd.randint(0, 3), rnd.randint(0, 3), rnd.randint(0, 3)
        sp = LevenshteinSearchParams(ms, mi, md, rnd.randint(0, ms + mi + md))
        a = sp.unpacked
        assert [(x.start, x.end, x.dist) for x in ref_lp(p, t, sp)] == \
            [r[:3] for r in oracle.generic_lp_raw(p, t, *a)], (p, t, a)
        if len(p) // (a[3] + 1) >= 1:
            assert [(x.start, x.end, x.dist) for x in ref_ng(p, t, sp)] == \
                [r[:3] for r in oracle.generic_ngrams_raw(p, t, *a)], (p, t, a)


def test_linear_programming_fallbacks_raw(ref):
    """(f)3: the oracle's restatement of the linear-programming fallbacks against the reference's own functions
    (levenshtein.py:52-148, substitutions_only.py:82-136) — ordered emission lists, short and long patterns,
    budgets up to and beyond the pattern length."""
    from fuzzysearch.levenshtein import find_near_matches_levenshtein_linear_programming as ref_lev_lp
    from fuzzysearch.substitutions_only import find_near_matches_substitutions_lp as ref_subs_lp
    rnd = random.Random(9)
    n_lev = n_subs = 0
    for _ in range(4000):
        alpha = bytes(rnd.sample(range(65, 91), rnd.choice([2, 2, 3, 4])))
        t = bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, 50)))
        m = rnd.randint(1, 10)
        if rnd.random() < 0.5 and len(t) >= m:
            st = rnd.randint(0, len(t) - m)
            p = bytearray(t[st:st + m])
            if rnd.random() < 0.5:
                p[rnd.randrange(m)] = rnd.choice(alpha)
            p = bytes(p)
        else:
            p = bytes(rnd.choice(alpha) for _ in range(m))
        k = rnd.randint(0, 4)
        exp = [(x.start, x.end, x.dist) for x in ref_lev_lp(p, t, k)]
        assert [r[:3] for r in oracle.lev_lp_raw(p, t, k)] == exp, (p, t, k)
        n_lev += bool(exp)
        exp = [(x.start, x.end, x.dist) for x in ref_subs_lp(p, t, k)]
        assert [r[:3] for r in oracle.subs_lp_raw(p, t, k)] == exp, (p, t, k)
        n_subs += bool(exp)
    assert n_lev > 1500 and n_subs > 1000


def test_api_consolidated_tie_aware(ref):
    """find_near_matches(max_l_dist=k) == consolidate(raw) up to hash-seed dependent ties."""
    from tests import golden_io
    rnd = random.Random(5)
    strict = total = 0
    for _ in range(3000):
        p, t, k = _case(rnd, max_m=16)
        if len(p) // (k + 1) < 3:
            continue
        exp = [(x.start, x.end, x.dist) for x in ref.find_near_matches(p, t, max_l_dist=k)]
        raw = oracle.lev_ngrams_raw(p, t, k)
        got = oracle.consolidate(raw)
        assert golden_io.equal_modulo_ties(got, exp, raw), (p, t, k, got, exp)
        strict += got == exp
        total += 1
    assert total > 500 and strict >= 0.8 * total


def test_reference_unit_tests_pass_with_all_natives():
    """The oracle build of the reference is sane: its own suite passes (1 known error in dead code)."""
    env = dict(os.environ, PYTHONHASHSEED="0")
    code = (
        "import sys, unittest; sys.path.insert(0, %r); sys.path.insert(0, '/root/reference');"
        "from oracle import ref_loader; ref_loader.load_reference_package();"
        "s = unittest.defaultTestLoader.discover('/root/reference/tests', top_level_dir='/root/reference');"
        "r = unittest.TextTestRunner(verbosity=0, stream=open('/dev/null', 'w')).run(s);"
        "print(r.testsRun, len(r.failures), len(r.errors))"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, "-c", code], env=env, cwd="/tmp").decode().split()
    assert int(out[0]) > 600 and int(out[1]) == 0 and int(out[2]) <= 1, out
