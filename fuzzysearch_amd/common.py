"""Result record, search parameters and the strategy-class contract of the drop-in API.

Mirrors (behaviour, not code) src/fuzzysearch/common.py of the reference:
  Match                     common.py:15-32   attrs class, frozen + slots, eq/hash/order on
                                              (start, end, dist); ``matched`` excluded; validated.
                                              Here: a C type with the same behaviour (csrc/_fzmatch.c:
                                              start / end / dist as C integers inside the instance — a
                                              match is one allocation next to its ``matched`` slice),
                                              carrying the attrs field list, so attr.fields / evolve /
                                              asdict work; the attrs class itself without the extension
  LevenshteinSearchParams   common.py:35-116  validation (TypeError / ValueError) + normalisation
  FuzzySearchBase           common.py:192-209 search / consolidate_matches /
                                              extra_items_for_chunked_search
  consolidate_overlapping_matches, group_matches, get_best_match_in_group   common.py:145-189

Consolidation runs in libfzhip's host code (fz_consolidate / fz_group_best): sort + sweep instead
of the reference's O(M * groups) Python loop, with a deterministic tie-break (smallest start)
where the reference's answer depends on PYTHONHASHSEED (SURVEY.md trap 3).
"""
import attr

from . import _native

__all__ = [
    'Match', 'RawMatches', 'matches_from_rows', 'LevenshteinSearchParams', 'FuzzySearchBase',
    'group_matches', 'get_best_match_in_group', 'consolidate_overlapping_matches',
    'count_differences_with_maximum',
]

_UNLIMITED = 1 << 29


@attr.s(frozen=True, slots=True)
class Match(object):
    start = attr.ib(type=int, eq=True, hash=True)
    end = attr.ib(type=int, eq=True, hash=True)
    dist = attr.ib(type=int, eq=True, hash=True)
    matched = attr.ib(eq=False, hash=False)

    if __debug__:
        def __attrs_post_init__(self):
            if not (isinstance(self.start, int) and self.start >= 0):
                raise ValueError('start must be a non-negative integer')
            if not (isinstance(self.end, int) and self.end >= self.start):
                raise ValueError('end must be an integer no smaller than start')
            if not (isinstance(self.dist, int) and self.dist >= 0):
                raise ValueError('dist must be a non-negative integer')
            if self.matched is None:
                raise ValueError('matched must be supplied')


_AttrsMatch = Match                         # the reference's class as it stands (tests hold the C type against it)
try:                                        # csrc/_fzmatch.c, built by fuzzysearch_amd.build
    from . import _fzmatch
    if not hasattr(_fzmatch, 'Match'):      # a stale build of an older source
        _fzmatch = None
except ImportError:                         # not built: the attrs class, filled by the Python loop below (~3x slower)
    _fzmatch = None
if _fzmatch is not None:
    Match = _fzmatch.Match
    Match.__attrs_attrs__ = _AttrsMatch.__attrs_attrs__
_SET_START, _SET_END, _SET_DIST, _SET_MATCHED = (_AttrsMatch.start.__set__, _AttrsMatch.end.__set__, _AttrsMatch.dist.__set__,
                                                 _AttrsMatch.matched.__set__)


def _is_limit(x):
    return x is None or (isinstance(x, int) and x >= 0)


@attr.s(frozen=True, slots=True, init=False)
class LevenshteinSearchParams(object):
    """(max_substitutions, max_insertions, max_deletions, max_l_dist), validated and normalised."""
    max_substitutions = attr.ib(default=None)
    max_insertions = attr.ib(default=None)
    max_deletions = attr.ib(default=None)
    max_l_dist = attr.ib(default=None)

    def __init__(self, max_substitutions=None, max_insertions=None, max_deletions=None, max_l_dist=None):
        limits = (max_substitutions, max_insertions, max_deletions)
        if not all(_is_limit(x) for x in limits + (max_l_dist,)):
            raise TypeError("All limits must be positive integers or None.")
        if max_l_dist is None:
            given = [x is not None for x in limits]
            if not any(given):
                raise ValueError('No limitations given!')
            for ok, what in zip(given, ('substitutions', 'insertions', 'deletions')):
                if not ok:
                    raise ValueError('# %s must be limited!' % what)
        total = sum(_UNLIMITED if x is None else x for x in limits)
        if max_l_dist is None:
            norm = limits + (total,)
        else:
            norm = tuple(max_l_dist if x is None else min(x, max_l_dist) for x in limits) \
                + (min(max_l_dist, total),)
        for name, value in zip(('max_substitutions', 'max_insertions', 'max_deletions', 'max_l_dist'), norm):
            object.__setattr__(self, name, value)

    @property
    def unpacked(self):
        return (self.max_substitutions, self.max_insertions, self.max_deletions, self.max_l_dist)


def count_differences_with_maximum(sequence1, sequence2, max_differences):
    """common.py:119-142 / _common.c:115-173 (count_differences_with_maximum_byteslike): min(number of positions
    that differ, max_differences).  Bytes-like inputs of equal length are compared in one vectorised pass (the
    reference's native takes the same inputs and raises the same ValueError on unequal lengths); everything else
    item by item with the early exit.  Host side: the device paths count mismatches in their kernels."""
    try:
        a, b = memoryview(sequence1), memoryview(sequence2)
        simple = a.itemsize == 1 and b.itemsize == 1 and a.ndim == 1 and b.ndim == 1 and a.c_contiguous and b.c_contiguous
    except TypeError:
        simple = False
    if simple:
        if a.nbytes != b.nbytes:
            raise ValueError('The lengths of the given sequences must be equal.')
        import numpy as np
        n = int(np.count_nonzero(np.frombuffer(a, dtype=np.uint8) != np.frombuffer(b, dtype=np.uint8)))
        return min(n, max_differences) if max_differences >= 0 else n
    n_different = 0
    for x, y in zip(sequence1, sequence2):
        if x != y:
            n_different += 1
            if n_different == max_differences:
                break
    return n_different


class RawMatches(object):
    """A raw match stream as it left the C-ABI: an fz_match structured array (start, end, dist, block) plus
    the sequence the indices refer to.  Behaves like a read-only list of Match objects, but the objects —
    and their ``matched`` slices — are only built when somebody looks (common.py:15-32 builds one attrs
    object per raw match; at 2e5 raw matches that costs more than the search).  Consolidation works on
    the array and materialises the survivors only."""
    __slots__ = ('array', 'sequence', 'offset', '_list')

    def __init__(self, array, sequence, offset=0):
        self.array = array
        self.sequence = sequence
        self.offset = offset                # added to start / end (file API: chunk offsets are already global -> 0)
        self._list = None

    def _make(self, rows):
        # The Python fill, for rows that are not a C-contiguous array (and for everything without the extension).
        seq, off = self.sequence, self.offset
        if Match is not _AttrsMatch:                   # the C type: its constructor is the cheap path
            return [Match(s + off, e + off, d, seq[s:e]) for (s, e, d, _g) in rows]
        # The attrs class: rows from the C-ABI satisfy Match's invariants by construction (0 <= start <= end, dist >= 0),
        # so the objects are filled through the slot descriptors: 0.25 us each instead of 0.6 us through the attrs
        # __init__ + validation.
        new, cls = object.__new__, Match
        out = []
        for (s, e, d, _g) in rows:
            m = new(cls)
            _SET_START(m, s + off)
            _SET_END(m, e + off)
            _SET_DIST(m, d)
            _SET_MATCHED(m, seq[s:e])
            out.append(m)
        return out

    def _make_from_array(self, array):
        if _fzmatch is not None and array.flags.c_contiguous:
            return _fzmatch.make_matches(array, self.sequence, self.offset)
        return self._make(array.tolist())

    def materialize(self):
        if self._list is None:
            self._list = self._make_from_array(self.array)
        return self._list

    def subset(self, array):
        """Match objects for another fz_match array over the same sequence (e.g. the consolidated one)."""
        return self._make_from_array(array)

    def __len__(self):
        return len(self.array)

    def __iter__(self):
        return iter(self.materialize())

    def __getitem__(self, item):
        return self.materialize()[item]

    def __eq__(self, other):
        return self.materialize() == (other.materialize() if isinstance(other, RawMatches) else other)

    def __ne__(self, other):
        return not self == other

    def __repr__(self):
        return 'RawMatches(%r)' % (self.materialize(),)


def matches_from_rows(rows, sequence):
    """_native.OwnedRows (the result buffer of a C-ABI call) -> list of Match over `sequence`, built in C straight from
    the buffer (csrc/_fzmatch.c: make_matches_at); the buffer is released.  Falls back to the array path without the
    extension."""
    try:
        if _fzmatch is not None:
            return _fzmatch.make_matches_at(rows.address, rows.n, sequence, 0)
        return RawMatches(rows.to_array(), sequence).materialize()
    finally:
        rows.release()


def _rows_of(matches):
    return [(m.start, m.end, m.dist) for m in matches]


def group_matches(matches):
    """-> list of sets of overlapping matches, in the reference's group-list order."""
    matches = list(matches)
    groups = []   # [start, end, set]
    for match in matches:
        hit = [g for g in groups if not (match.end <= g[0] or match.start >= g[1])]
        if not hit:
            groups.append([match.start, match.end, {match}])
        elif len(hit) == 1:
            g = hit[0]
            g[0], g[1] = min(g[0], match.start), max(g[1], match.end)
            g[2].add(match)
        else:
            merged = [match.start, match.end, {match}]
            for g in hit:
                merged[0], merged[1] = min(merged[0], g[0]), max(merged[1], g[1])
                merged[2] |= g[2]
            groups = [g for g in groups if not any(g is h for h in hit)]
            groups.append(merged)
    return [g[2] for g in groups]


def get_best_match_in_group(group):
    """Longest of the smallest-distance matches; ties -> smallest start (deterministic)."""
    return min(group, key=lambda m: (m.dist, -(m.end - m.start), m.start))


def _reduce(matches, array_fn):
    """Run a C-side reduction (consolidation / best of groups) and hand back Match objects.  A RawMatches
    input never materialises the raw stream: only the survivors become objects."""
    if isinstance(matches, RawMatches):
        if len(matches) == 0:
            return []
        if matches.offset:
            raise ValueError('offset streams must be rebased first')
        return matches.subset(array_fn(matches.array))
    matches = list(matches)
    if not matches:
        return []
    by_key = {}
    for m in matches:
        by_key.setdefault((m.start, m.end, m.dist), m)
    best = array_fn(_rows_of(matches)).tolist()
    return [by_key[(s, e, d)] for (s, e, d, _b) in best]


def consolidate_overlapping_matches(matches):
    """One best match per group of overlapping matches, sorted (common.py:185-189)."""
    return _reduce(matches, _native.consolidate_array)


def best_of_groups_in_discovery_order(matches):
    """[get_best_match_in_group(g) for g in group_matches(matches)] (substitutions_only.py:279-282)."""
    return _reduce(matches, _native.group_best_array)


class FuzzySearchBase(object):
    """Strategy-class contract (common.py:192-209)."""

    @classmethod
    def search(cls, subsequence, sequence, search_params):
        raise NotImplementedError

    @classmethod
    def consolidate_matches(cls, matches):
        if isinstance(matches, RawMatches):
            return matches.materialize()
        try:
            len(matches)
        except TypeError:
            return list(matches)
        else:
            return matches

    @classmethod
    def extra_items_for_chunked_search(cls, subsequence, search_params):
        raise NotImplementedError
