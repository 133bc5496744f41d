"""fuzzysearch_amd — MI355X-native fuzzy substring search, a drop-in for taleinat/fuzzysearch.

Same public surface as the reference (src/fuzzysearch/__init__.py:18-22): ``find_near_matches``,
``find_near_matches_in_file`` and ``Match``.  The Levenshtein n-gram hot path (and its
substitutions-only / exact forms) runs as hand-written HIP on gfx950 through the C-ABI of
include/fzhip.h; this package is plain Python + ctypes (no PyTorch) and has NO CPU fallback.

>>> find_near_matches(b'PATTERN', b'---PATERN---', max_l_dist=1)
[Match(start=3, end=9, dist=1, matched=b'PATERN')]

Extra, MI355X-specific surface: ``resident(sequence)`` uploads a long sequence to HBM once so that
many patterns can be searched without re-crossing PCIe.
"""
import io

from .common import Match, LevenshteinSearchParams
from .engine import DeviceSequence, resident, cache_info, cache_clear, residency_cache
from ._native import UnsupportedSearch
from .generic_search import GenericSearch
from .levenshtein import LevenshteinSearch
from .search_exact import ExactSearch
from .substitutions_only import SubstitutionsOnlySearch
from . import _file_stream

__version__ = '0.1.0'

__all__ = [
    'find_near_matches',
    'find_near_matches_in_file',
    'Match',
    'resident',
    'cache_info',
    'cache_clear',
    'UnsupportedSearch',
]


def choose_search_class(search_params):
    """Strategy choice, rules of src/fuzzysearch/__init__.py:60-83."""
    max_subs, max_ins, max_dels, max_l = search_params.unpacked
    if max_l == 0:
        return ExactSearch
    if max_ins == 0 and max_dels == 0:
        return SubstitutionsOnlySearch
    unlimited = 1 << 29
    if max_l <= min(unlimited if x is None else x for x in (max_subs, max_ins, max_dels)):
        return LevenshteinSearch
    return GenericSearch


def find_near_matches(subsequence, sequence,
                      max_substitutions=None,
                      max_insertions=None,
                      max_deletions=None,
                      max_l_dist=None):
    """search for near-matches of subsequence in sequence

    Limits (relative to the subsequence): maximum substitutions, insertions, deletions and their
    total (the Levenshtein distance).  ``sequence`` may also be a ``resident()`` handle.

    Every route runs on the GPU; there is no CPU fallback.  What the engine does not support raises
    ``UnsupportedSearch`` (a ``NotImplementedError``) before anything is searched: a subsequence of more
    than 65 535 items, ``max_l_dist`` above 1 023 (above 255 for searches with separate substitution /
    insertion / deletion limits and for the short-pattern routes), more than 255 distinct symbols in a
    subsequence that is neither bytes nor latin-1 text, and generic searches whose candidate sets outgrow
    2**18 entries.  The reference accepts all of these: catch the exception to route such a call there.

    Residency: a ``bytes`` or ``str`` sequence of 64 KiB or more stays in device memory after the call, so the next
    query against the SAME object uploads nothing.  The cache lets go of a sequence as soon as the caller has (the
    entry of an object that nothing else references is dropped at the next call, with its device memory), holds at
    most 16 sequences and a quarter of the device memory that was free at its first use (at most 8 GiB;
    ``FUZZYSEARCH_HIP_RESIDENT_CACHE=<bytes, K / M / G>`` sets the budget, ``0`` switches the cache off);
    ``fuzzysearch_amd.cache_info()`` shows what it holds, ``cache_clear()`` empties it.
    """
    search_params = LevenshteinSearchParams(max_substitutions, max_insertions,
                                            max_deletions, max_l_dist)
    search_class = choose_search_class(search_params)
    fused = getattr(search_class, 'search_consolidated', None)
    if fused is not None:                          # search + consolidation in one device round trip (generic n-gram route)
        result = fused(subsequence, sequence, search_params)
        if result is not None:
            return result
    matches = search_class.search(subsequence, sequence, search_params)
    return search_class.consolidate_matches(matches)


def find_near_matches_in_file(subsequence, sequence_file,
                              max_substitutions=None,
                              max_insertions=None,
                              max_deletions=None,
                              max_l_dist=None,
                              _chunk_size=2**20):
    """search for near-matches of subsequence in a file (src/fuzzysearch/__init__.py:86-200).

    Reproduces the reference's chunk geometry exactly (``_chunk_size`` windows overlapping by
    ``len(subsequence) - 1 + extra`` items, every chunk searched as an independent sequence, one
    global consolidation at the end) because the result depends on it.

    Regular files, in-memory binary files and text files go through the streaming pipeline
    (fz_stream): many chunks cross PCIe as one batch from pinned, double-buffered memory and are
    searched by one launch with per-chunk clamps.  Everything else (unseekable streams, patterns on the
    reference's linear-programming routes, chunks shorter than twice the overlap) is searched chunk by
    chunk, exactly as the reference does it.
    """
    search_params = LevenshteinSearchParams(max_substitutions, max_insertions,
                                            max_deletions, max_l_dist)
    search_class = choose_search_class(search_params)
    if not len(subsequence):
        raise ValueError('subsequence must not be empty')
    binary = 'b' in getattr(sequence_file, 'mode', '') or isinstance(sequence_file, io.RawIOBase)
    keep = len(subsequence) - 1 + search_class.extra_items_for_chunked_search(subsequence, search_params)
    plan = _file_stream.plan(search_class, subsequence, search_params, _chunk_size, keep, binary, sequence_file)
    if plan is not None:
        try:
            return _file_stream.run(plan, search_class, subsequence, sequence_file)
        except _file_stream.Unsupported:
            pass                                   # nothing has been read yet: take the per-chunk path
    with residency_cache().bypass():               # every chunk is a new object that is searched once: nothing to keep resident
        if binary:
            matches = _search_binary_file(subsequence, sequence_file, search_params, search_class, _chunk_size, keep)
        else:
            matches = _search_text_file(subsequence, sequence_file, search_params, search_class, _chunk_size, keep)
    return search_class.consolidate_matches(matches)


def _search_binary_file(subsequence, sequence_file, search_params, search_class, chunk_size, keep):
    pattern = bytearray(subsequence)
    buf = bytearray(chunk_size)
    view = memoryview(buf)
    out = []
    offset = 0
    n_read = sequence_file.readinto(view)
    chunk_len = n_read
    while n_read:
        chunk = buf if chunk_len == chunk_size else buf[:chunk_len]
        for match in search_class.search(pattern, chunk, search_params):
            out.append(Match(match.start + offset, match.end + offset, match.dist, match.matched))
        n_keep = min(keep, chunk_len) if keep > 0 else 0
        if n_keep:
            view[:n_keep] = bytes(view[chunk_len - n_keep:chunk_len])
        offset += chunk_len - n_keep
        n_read = sequence_file.readinto(view[n_keep:])
        chunk_len = n_keep + n_read
    return out


def _search_text_file(subsequence, sequence_file, search_params, search_class, chunk_size, keep):
    out = []
    offset = 0
    chunk = sequence_file.read(chunk_size)
    while chunk:
        for match in search_class.search(subsequence, chunk, search_params):
            out.append(Match(match.start + offset, match.end + offset, match.dist, match.matched))
        n_keep = min(keep, len(chunk))
        offset += len(chunk) - n_keep
        if n_keep:
            chunk = chunk[-n_keep:] + sequence_file.read(chunk_size)
            if len(chunk) == n_keep:
                break
        else:
            chunk = sequence_file.read(chunk_size)
    return out
