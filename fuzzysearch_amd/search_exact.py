"""Exact search on the GPU: the n-gram filter kernel with one block of length len(subsequence).

Mirrors src/fuzzysearch/search_exact.py: ``search_exact`` (:59-77 -> search_exact_byteslike,
_common.c:5-112) and ``ExactSearch`` (:80-89).
"""
from .common import FuzzySearchBase, RawMatches
from .engine import prepare

__all__ = ['search_exact', 'ExactSearch']


def search_exact(subsequence, sequence, start_index=0, end_index=None):
    """All (overlapping) occurrences of subsequence in sequence[start_index:end_index], ascending."""
    if not len(subsequence):
        raise ValueError('subsequence must not be empty')
    n = len(sequence)
    if end_index is None:
        end_index = n
    start_index = max(0, min(start_index, n))                 # clamp, search_exact.py:70-71
    end_index = max(start_index, min(end_index, n))
    pr = prepare(subsequence, sequence)
    try:
        return pr.engine.search_exact(pr.handle, pr.pattern, start_index, end_index)
    finally:
        pr.release()


def exact_raw(subsequence, sequence):
    """Every occurrence as a RawMatches stream (start, start + m, 0): the indices stay a numpy array, Match objects
    are only built when somebody looks at them (search_exact.py:83-89 builds one per hit)."""
    import numpy as np
    from . import _native
    if not len(subsequence):
        raise ValueError('subsequence must not be empty')
    pr = prepare(subsequence, sequence)
    try:
        idx = pr.engine.search_exact(pr.handle, pr.pattern, as_array=True)
    finally:
        pr.release()
    raw = np.empty(len(idx), dtype=_native._match_dtype())
    raw["start"] = idx
    raw["end"] = idx + len(subsequence)
    raw["dist"] = 0
    raw["block"] = -1
    return RawMatches(raw, pr.original)


class ExactSearch(FuzzySearchBase):
    @classmethod
    def search(cls, subsequence, sequence, search_params):
        return exact_raw(subsequence, sequence)

    @classmethod
    def extra_items_for_chunked_search(cls, subsequence, search_params):
        return 0
