"""Generic (separate substitution / insertion / deletion limits) search.

Mirrors src/fuzzysearch/generic_search.py:25-54, :198-237, :256-273.
"""
from .common import FuzzySearchBase, Match, RawMatches, consolidate_overlapping_matches, matches_from_rows
from .engine import prepare
from .search_exact import exact_raw

__all__ = ['find_near_matches_generic', 'find_near_matches_generic_ngrams',
           'find_near_matches_generic_linear_programming', 'has_near_match_generic_ngrams',
           'GenericSearch']


def find_near_matches_generic_ngrams(subsequence, sequence, search_params):
    return raw_generic_ngrams(subsequence, sequence, search_params).materialize()


def raw_generic_ngrams(subsequence, sequence, search_params):
    if not len(subsequence):
        raise ValueError('Given subsequence is empty!')
    max_subs, max_ins, max_dels, max_l = search_params.unpacked
    if len(subsequence) // (max_l + 1) == 0:
        raise ValueError('the subsequence length must be greater than max_l_dist')
    pr = prepare(subsequence, sequence)
    try:
        raw = pr.engine.generic_ngrams(pr.handle, pr.pattern, max_subs, max_ins, max_dels, max_l, as_array=True)
    finally:
        pr.release()
    return RawMatches(raw, pr.original)


def find_near_matches_generic(subsequence, sequence, search_params):
    res = raw_generic(subsequence, sequence, search_params)
    return res.materialize() if isinstance(res, RawMatches) else res


def raw_generic(subsequence, sequence, search_params):
    if not len(subsequence):
        raise ValueError('Given subsequence is empty!')
    m = len(subsequence)
    if search_params.max_l_dist == 0:
        return exact_raw(subsequence, sequence)
    if m // (search_params.max_l_dist + 1) >= 3:
        return raw_generic_ngrams(subsequence, sequence, search_params)
    return raw_generic_lp(subsequence, sequence, search_params)


def find_near_matches_generic_linear_programming(subsequence, sequence, search_params):
    """generic_search.py:57-177 over the whole sequence, on the GPU tiled by start position."""
    return raw_generic_lp(subsequence, sequence, search_params).materialize()


def raw_generic_lp(subsequence, sequence, search_params):
    if not len(subsequence):
        raise ValueError('Given subsequence is empty!')
    unlimited = 1 << 29
    max_subs, max_ins, max_dels, max_l = (unlimited if x is None else x for x in search_params.unpacked)
    pr = prepare(subsequence, sequence)
    try:
        # max_l goes through unclamped: beyond the kernels' limit the library answers FZ_EUNSUPPORTED
        # (NotImplementedError) instead of silently dropping matches; the three per-kind limits may be
        # clamped, each is <= max_l anyway
        raw = pr.engine.generic_lp(pr.handle, pr.pattern, min(max_subs, unlimited), min(max_ins, unlimited),
                                   min(max_dels, unlimited), min(max_l, (1 << 32) - 1), as_array=True)
    finally:
        pr.release()
    return RawMatches(raw, pr.original)


def has_near_match_generic_ngrams(subsequence, sequence, search_params):
    """generic_search.py:240-253: is there any match?  A flag-only search (fz_generic_ngrams_any): no rows are
    ordered, copied or turned into Match objects, and automaton work that starts after the first match is skipped."""
    if not len(subsequence):
        raise ValueError('Given subsequence is empty!')
    max_subs, max_ins, max_dels, max_l = search_params.unpacked
    if len(subsequence) // (max_l + 1) == 0:
        raise ValueError('the subsequence length must be greater than max_l_dist')
    pr = prepare(subsequence, sequence)
    try:
        return pr.engine.generic_ngrams_any(pr.handle, pr.pattern, max_subs, max_ins, max_dels, max_l)
    finally:
        pr.release()


class GenericSearch(FuzzySearchBase):
    @classmethod
    def search(cls, subsequence, sequence, search_params):
        return raw_generic(subsequence, sequence, search_params)

    @classmethod
    def search_consolidated(cls, subsequence, sequence, search_params):
        """search() + consolidate_matches() in one device round trip for the n-gram route (what find_near_matches
        needs: generic_search.py:198-237 + common.py:185-189): the automaton kernel folds every hit's matches into
        (hull, best match) pairs, so a few thousand pairs cross PCIe instead of every raw match
        (fz_generic_ngrams_consolidated).  -> list of Match, or None when the call takes another route."""
        if not len(subsequence):
            raise ValueError('Given subsequence is empty!')
        max_subs, max_ins, max_dels, max_l = search_params.unpacked
        if max_l == 0 or len(subsequence) // (max_l + 1) < 3:
            return None
        pr = prepare(subsequence, sequence)
        try:
            rows = pr.engine.rows_call(pr.engine._lib.fz_generic_ngrams_consolidated, pr.handle, pr.pattern, max_subs, max_ins,
                                       max_dels, max_l)
        finally:
            pr.release()
        return matches_from_rows(rows, pr.original)

    @classmethod
    def consolidate_matches(cls, matches):
        return consolidate_overlapping_matches(matches)

    @classmethod
    def extra_items_for_chunked_search(cls, subsequence, search_params):
        return max(x for x in (search_params.max_l_dist, search_params.max_insertions) if x is not None)
