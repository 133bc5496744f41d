"""Substitutions-only search (mirrors src/fuzzysearch/substitutions_only.py:37-63, :236-301).

SURVEY.md trap 5: for bytes-like input the reference returns the best match of every overlap
group in group-creation order (C path + group_matches, :258-282); for ``str`` it returns every
Hamming <= k window sorted by start (pure-Python path, :160-167).  Both are reproduced from the
same GPU raw stream (fz_subs_ngrams).
"""
from .common import FuzzySearchBase, Match, RawMatches, best_of_groups_in_discovery_order, matches_from_rows
from .engine import DeviceSequence, is_byteslike, prepare
from .search_exact import exact_raw, search_exact

__all__ = ['find_near_matches_substitutions', 'find_near_matches_substitutions_ngrams',
           'find_near_matches_substitutions_lp',
           'has_near_match_substitutions', 'has_near_match_substitutions_ngrams',
           'has_near_match_substitutions_lp',
           'SubstitutionsOnlySearch']


def _check_arguments(subsequence, sequence, max_substitutions):
    if not len(subsequence):
        raise ValueError('Given subsequence is empty!')
    if max_substitutions is None or max_substitutions < 0:
        raise ValueError('Maximum number of substitutions must be >= 0!')


def find_near_matches_substitutions_ngrams(subsequence, sequence, max_substitutions):
    _check_arguments(subsequence, sequence, max_substitutions)
    m = len(subsequence)
    pr = prepare(subsequence, sequence)
    try:
        if m // (max_substitutions + 1) == 0:
            # n-gram length 0.  The reference's C path (bytes-like input) answers every window
            # (_substitutions_only_ngrams_template.h:76-88: `max_substitutions >= len` -> all starts), its
            # pure-Python path (str) raises (substitutions_only.py:239-242).
            if not pr.byteslike:
                raise ValueError("The subsequence's length must be greater than max_substitutions!")
            raw = pr.engine.subs_lp(pr.handle, pr.pattern, max_substitutions, as_array=True)
        else:
            raw = pr.engine.subs_ngrams(pr.handle, pr.pattern, max_substitutions, as_array=True)
    finally:
        pr.release()
    return _finish_ngrams(raw, pr.original, pr.byteslike)


def _finish_ngrams(raw, seq, byteslike):
    """Raw stream -> what the reference's wrapper returns: bytes-like input -> best of every overlap group in
    group-creation order (:266-282); str -> every window once, sorted by start (:160-167)."""
    if byteslike:
        return best_of_groups_in_discovery_order(RawMatches(raw, seq))
    import numpy as np
    _starts, first = np.unique(raw["start"], return_index=True)     # sorted by start, first occurrence of each
    return RawMatches(raw[first], seq).materialize()


def find_near_matches_substitutions(subsequence, sequence, max_substitutions):
    _check_arguments(subsequence, sequence, max_substitutions)
    m = len(subsequence)
    if max_substitutions == 0:
        return exact_raw(subsequence, sequence).materialize()
    if m // (max_substitutions + 1) >= 3:
        return find_near_matches_substitutions_ngrams(subsequence, sequence, max_substitutions)
    return find_near_matches_substitutions_lp(subsequence, sequence, max_substitutions)


def find_near_matches_substitutions_lp(subsequence, sequence, max_substitutions):
    """substitutions_only.py:65-136: every window with at most max_substitutions mismatches,
    ascending start (fz_subs_lp: a Hamming kernel without the n-gram filter)."""
    _check_arguments(subsequence, sequence, max_substitutions)
    pr = prepare(subsequence, sequence)
    try:
        raw = pr.engine.subs_lp(pr.handle, pr.pattern, max_substitutions, as_array=True)
    finally:
        pr.release()
    return RawMatches(raw, pr.original).materialize()


def _any_raw(call, subsequence, sequence, max_substitutions):
    """Flag-only searches (fz_subs_ngrams_any / fz_subs_lp_any): nothing is ordered or copied, and scan work that
    starts after the first match has been counted is skipped (the reference stops at its first match, :218-233)."""
    pr = prepare(subsequence, sequence)
    try:
        return call(pr.engine)(pr.handle, pr.pattern, max_substitutions)
    finally:
        pr.release()


def has_near_match_substitutions_ngrams(subsequence, sequence, max_substitutions):
    """substitutions_only.py:218-233: does any window have at most max_substitutions mismatches?
    (The reference stops at the first one; here workgroups that start after the first match skip their tiles.)"""
    _check_arguments(subsequence, sequence, max_substitutions)
    if len(subsequence) // (max_substitutions + 1) == 0:
        if isinstance(subsequence, str):
            raise ValueError("The subsequence's length must be greater than max_substitutions!")
        return len(sequence) >= len(subsequence)          # template.h:76-88: every window matches
    return _any_raw(lambda eng: eng.subs_ngrams_any, subsequence, sequence, max_substitutions)


def has_near_match_substitutions_lp(subsequence, sequence, max_substitutions):
    """substitutions_only.py:139-145."""
    _check_arguments(subsequence, sequence, max_substitutions)
    return _any_raw(lambda eng: eng.subs_lp_any, subsequence, sequence, max_substitutions)


def has_near_match_substitutions(subsequence, sequence, max_substitutions):
    """substitutions_only.py:18-34 (same dispatch rule as find_near_matches_substitutions)."""
    _check_arguments(subsequence, sequence, max_substitutions)
    if max_substitutions == 0:
        return len(search_exact(subsequence, sequence)) > 0
    if len(subsequence) // (max_substitutions + 1) >= 3:
        return has_near_match_substitutions_ngrams(subsequence, sequence, max_substitutions)
    return has_near_match_substitutions_lp(subsequence, sequence, max_substitutions)


class SubstitutionsOnlySearch(FuzzySearchBase):
    @classmethod
    def search(cls, subsequence, sequence, search_params):
        k = min(x for x in (search_params.max_l_dist, search_params.max_substitutions) if x is not None)
        return find_near_matches_substitutions(subsequence, sequence, k)

    @classmethod
    def search_consolidated(cls, subsequence, sequence, search_params):
        """The n-gram route on bytes-like input in ONE C-ABI call (fz_subs_ngrams_best: search + best of every overlap
        group in group-list order, substitutions_only.py:258-282), Match objects built in C straight from the result
        buffer.  -> list of Match, or None for every other route (str input: every window sorted by start; exact;
        linear programming; n-gram length 0)."""
        k = min(x for x in (search_params.max_l_dist, search_params.max_substitutions) if x is not None)
        _check_arguments(subsequence, sequence, k)
        if k == 0 or len(subsequence) // (k + 1) < 3:
            return None
        if isinstance(sequence, DeviceSequence):
            if not sequence.byteslike:
                return None
        elif not (is_byteslike(sequence) and is_byteslike(subsequence)):
            return None                                    # (decided before anything is uploaded)
        pr = prepare(subsequence, sequence)
        try:
            rows = pr.engine.rows_call(pr.engine._lib.fz_subs_ngrams_best, pr.handle, pr.pattern, k)
        finally:
            pr.release()
        return matches_from_rows(rows, pr.original)

    @classmethod
    def extra_items_for_chunked_search(cls, subsequence, search_params):
        return 0
