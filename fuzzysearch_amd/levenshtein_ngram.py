"""Levenshtein n-gram search — the north-star hot path — on the GPU.

Mirrors ``find_near_matches_levenshtein_ngrams`` (src/fuzzysearch/levenshtein_ngram.py:159-198):
same arguments, same ValueError, same raw (un-consolidated, ordered) stream of Match objects.  One
fz_lev_ngrams call replaces the reference's G passes of search_exact_byteslike plus two
c_expand_* calls per n-gram hit.
"""
from .common import RawMatches
from .engine import prepare

__all__ = ['find_near_matches_levenshtein_ngrams']


def raw_levenshtein_ngrams(subsequence, sequence, max_l_dist):
    """-> RawMatches (the stream as an array; Match objects only on demand)."""
    m = len(subsequence)
    if m // (max_l_dist + 1) == 0:
        raise ValueError('the subsequence length must be greater than max_l_dist')
    pr = prepare(subsequence, sequence)
    try:
        raw = pr.engine.lev_ngrams(pr.handle, pr.pattern, max_l_dist, as_array=True)
    finally:
        pr.release()
    return RawMatches(raw, pr.original)


def find_near_matches_levenshtein_ngrams(subsequence, sequence, max_l_dist):
    return raw_levenshtein_ngrams(subsequence, sequence, max_l_dist).materialize()
