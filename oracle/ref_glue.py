"""CPU baseline driver over the REFERENCE's own native functions.  TEST INFRASTRUCTURE ONLY.

``/root/reference`` (and with it the reference's Python glue) does not exist on the GPU box, but
the reference's compiled natives in ``oracle/_ref`` travel there.  This module restates ONLY the
Python outer loop of ``find_near_matches_levenshtein_ngrams`` (levenshtein_ngram.py:159-198) and
``_expand`` (:8-19) and calls the reference's own ``search_exact_byteslike`` (_common.c:5-112) and
``c_expand_short`` / ``c_expand_long`` (_levenshtein_ngrams.pyx:9-154) for all the arithmetic — the
same native calls, in the same order, as the reference makes.  bench.py times it as
``cpu_baseline.kind = "reference"``.
"""
from . import ref_loader

_natives = None


def natives():
    global _natives
    if _natives is None:
        _natives = ref_loader.load_reference_natives()
    return _natives


def lev_ngrams_raw(p, t, k):
    """-> list of (start, end, dist) in reference emission order, single thread, GIL held."""
    nat = natives()
    search = nat["_common"].search_exact_byteslike
    exp_short = nat["_levenshtein_ngrams"].c_expand_short
    exp_long = nat["_levenshtein_ngrams"].c_expand_long

    def expand(sub, win, budget):                        # levenshtein_ngram.py:8-19
        if len(sub) > max(budget * 2, 10):
            return exp_long(sub, win, budget)
        return exp_short(sub, win, budget)

    m, n = len(p), len(t)
    L = m // (k + 1)
    if L == 0:
        raise ValueError('the subsequence length must be greater than max_l_dist')
    out = []
    for s in range(0, m - L + 1, L):
        e = s + L
        before_rev = p[:s][::-1]
        after = p[e:]
        lo = max(0, s - k)
        hi = min(n, n - m + e + k)
        lo = max(0, min(lo, n))
        hi = max(lo, min(hi, n))
        for idx in search(p[s:e], t, lo, hi):
            d_r, r = expand(after, t[idx + L: idx - s + m + k], k)
            if d_r is None:
                continue
            d_l, l = expand(before_rev, t[max(0, idx - s - (k - d_r)): idx][::-1], k - d_r)
            if d_l is None:
                continue
            out.append((idx - l, idx + L + r, d_l + d_r))
    return out
