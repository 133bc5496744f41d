"""Levenshtein search dispatcher (mirrors src/fuzzysearch/levenshtein.py:9-38, :151-164)."""
from .common import FuzzySearchBase, Match, RawMatches, consolidate_overlapping_matches, matches_from_rows
from .engine import prepare
from .levenshtein_ngram import raw_levenshtein_ngrams
from .search_exact import exact_raw

__all__ = ['find_near_matches_levenshtein', 'find_near_matches_levenshtein_linear_programming',
           'LevenshteinSearch']


def find_near_matches_levenshtein(subsequence, sequence, max_l_dist):
    res = raw_levenshtein(subsequence, sequence, max_l_dist)
    return res.materialize() if isinstance(res, RawMatches) else res


def raw_levenshtein(subsequence, sequence, max_l_dist):
    if not len(subsequence):
        raise ValueError('Given subsequence is empty!')
    if max_l_dist < 0:
        raise ValueError('Maximum Levenshtein distance must be >= 0!')
    m = len(subsequence)
    if max_l_dist == 0:
        return exact_raw(subsequence, sequence)
    if m // (max_l_dist + 1) >= 3:
        return raw_levenshtein_ngrams(subsequence, sequence, max_l_dist)
    return raw_levenshtein_lp(subsequence, sequence, max_l_dist)


def raw_levenshtein_lp(subsequence, sequence, max_l_dist):
    if not len(subsequence):
        raise ValueError('Given subsequence is empty!')
    pr = prepare(subsequence, sequence)
    try:
        raw = pr.engine.lev_lp(pr.handle, pr.pattern, max_l_dist, as_array=True)
    finally:
        pr.release()
    return RawMatches(raw, pr.original)


def find_near_matches_levenshtein_linear_programming(subsequence, sequence, max_l_dist):
    """levenshtein.py:52-148 — the candidate automaton for short patterns, run on the GPU tiled by
    start position (fz_lev_lp); same ordered list of matches as the reference yields."""
    return raw_levenshtein_lp(subsequence, sequence, max_l_dist).materialize()


class LevenshteinSearch(FuzzySearchBase):
    @classmethod
    def search(cls, subsequence, sequence, search_params):
        return raw_levenshtein(subsequence, sequence, search_params.max_l_dist)

    @classmethod
    def search_consolidated(cls, subsequence, sequence, search_params):
        """search() + consolidate_matches() in ONE C-ABI call for the n-gram route (fz_lev_ngrams_consolidated: the
        consolidation runs on the rows where they are) and Match objects built in C straight from the result buffer.
        -> list of Match, or None when the call takes another route (exact, linear programming)."""
        if not len(subsequence):
            raise ValueError('Given subsequence is empty!')
        k = search_params.max_l_dist
        if k == 0 or len(subsequence) // (k + 1) < 3:
            return None
        pr = prepare(subsequence, sequence)
        try:
            rows = pr.engine.rows_call(pr.engine._lib.fz_lev_ngrams_consolidated, pr.handle, pr.pattern, k)
        finally:
            pr.release()
        return matches_from_rows(rows, pr.original)

    @classmethod
    def consolidate_matches(cls, matches):
        return consolidate_overlapping_matches(matches)

    @classmethod
    def extra_items_for_chunked_search(cls, subsequence, search_params):
        return search_params.max_l_dist
