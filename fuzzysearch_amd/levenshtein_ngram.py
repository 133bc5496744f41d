"""Levenshtein n-gram search — the north-star hot path — on the GPU.

Mirrors ``find_near_matches_levenshtein_ngrams`` (src/fuzzysearch/levenshtein_ngram.py:159-198):
same arguments, same ValueError, same raw (un-consolidated, ordered) stream of Match objects.  One
fz_lev_ngrams call replaces the reference's G passes of search_exact_byteslike plus two
c_expand_* calls per n-gram hit.
"""
from .common import Match
from .engine import prepare

__all__ = ['find_near_matches_levenshtein_ngrams']


def find_near_matches_levenshtein_ngrams(subsequence, sequence, max_l_dist):
    m = len(subsequence)
    if m // (max_l_dist + 1) == 0:
        raise ValueError('the subsequence length must be greater than max_l_dist')
    pr = prepare(subsequence, sequence)
    try:
        raw = pr.engine.lev_ngrams(pr.handle, pr.pattern, max_l_dist)
    finally:
        pr.release()
    seq = pr.original
    return [Match(s, e, d, matched=seq[s:e]) for (s, e, d, _g) in raw]
