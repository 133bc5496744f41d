"""Host-side glue between Python sequences and the HIP engine.

* ``encode_pair`` turns (subsequence, sequence) of any type the reference accepts into two byte
  buffers with IDENTICAL index semantics, so that every (start, end, dist) the kernels produce is
  valid for the original objects:
    - bytes-like (bytes, bytearray, memoryview, numpy uint8 ...): zero-copy pass-through — the
      reference's "byteslike" native path (_c_ext_base.h:27-34);
    - ``str``: latin-1 encoding when every code point fits a byte (same indices), else
    - anything else (str with wide code points, list, tuple): symbol remapping.  All algorithms on
      this path only ever compare a subsequence item with a sequence item
      (levenshtein_ngram.py:55, _generic_search.pyx:100, search_exact.py:46), so mapping each
      distinct subsequence symbol to a code 1..255 and every other sequence symbol to 0 preserves
      every comparison.
* ``DeviceSequence`` keeps a sequence resident in HBM across queries (upload once, search often).
"""
import sys

import numpy as np

from . import _native
from ._native import UnsupportedSearch

__all__ = ['DeviceSequence', 'resident', 'encode_pair', 'is_byteslike', 'ResidencyCache', 'residency_cache']


_BIO_SEQ = []                                    # [Seq class or None] once looked up


def _bio_seq():
    """Bio.Seq.Seq when Biopython is installed (search_exact.py:13-19 registers it as a sequence type), else None.
    Looked up once: a failing import walks sys.path every time (0.15 ms — more than the Match objects of a search)."""
    if _BIO_SEQ and _BIO_SEQ[0] is None and 'Bio.Seq' in sys.modules:
        del _BIO_SEQ[:]                              # Biopython was imported (or installed) after the first look
    if not _BIO_SEQ:
        try:
            from Bio.Seq import Seq
            _BIO_SEQ.append(Seq)
        except ImportError:
            _BIO_SEQ.append(None)
    return _BIO_SEQ[0]


def _unwrap_bio(x):
    """A Biopython Seq -> its letters as bytes (indices are the same: one byte per letter); anything else unchanged.
    Such inputs take the reference's pure-Python semantics (no buffer protocol -> its natives refuse them), i.e. they
    are not "bytes-like" for the substitutions-only result form (SURVEY.md trap 5)."""
    Seq = _bio_seq()
    if Seq is not None and isinstance(x, Seq):
        return bytes(x), True
    return x, False


def is_byteslike(x):
    if isinstance(x, (bytes, bytearray)):
        return True
    if isinstance(x, (str, list, tuple)):
        return False
    try:
        mv = memoryview(x)
    except TypeError:
        return False
    return mv.itemsize == 1 and mv.ndim == 1 and mv.c_contiguous


def _remap_str(subsequence, sequence):
    p = np.frombuffer(subsequence.encode('utf-32-le'), dtype=np.uint32)
    t = np.frombuffer(sequence.encode('utf-32-le'), dtype=np.uint32)
    symbols = np.unique(p)
    if len(symbols) > 255:
        raise UnsupportedSearch('subsequences with more than 255 distinct symbols are not supported')

    def code(a):
        if len(a) == 0:
            return b''
        pos = np.minimum(np.searchsorted(symbols, a), len(symbols) - 1)
        return np.where(symbols[pos] == a, pos + 1, 0).astype(np.uint8).tobytes()
    return code(p), code(t)


def _remap_items(subsequence, sequence):
    """list / tuple inputs: every distinct subsequence item -> a code 1..255, every other sequence item -> 0 (all the
    algorithms on this path only compare a subsequence item with a sequence item).  Integer items (the common case of
    non-text sequences) are coded by one vectorised numpy pass; anything else by a dict lookup per item that runs inside
    `map` (no Python-level loop body)."""
    try:
        p = np.asarray(subsequence)
        t = np.asarray(sequence)
        if p.dtype.kind in 'iub' and t.dtype.kind in 'iub' and p.ndim == 1 and t.ndim == 1:
            # one common integer type that holds every value of both arrays WITHOUT wrapping (uint64 values >= 2**63 do not
            # fit int64 and negative values do not fit uint64: casting would code 2**64 - 1 as -1 and report a match the
            # reference's == does not see) — otherwise the general form below
            big_p = p.dtype == np.uint64 and len(p) and int(p.max()) > np.iinfo(np.int64).max
            big_t = t.dtype == np.uint64 and len(t) and int(t.max()) > np.iinfo(np.int64).max
            if big_p or big_t:
                if (p.dtype.kind == 'i' and len(p) and int(p.min()) < 0) or (t.dtype.kind == 'i' and len(t) and int(t.min()) < 0):
                    raise OverflowError('no common integer type')
                p, t = p.astype(np.uint64, copy=False), t.astype(np.uint64, copy=False)
            else:
                p, t = p.astype(np.int64, copy=False), t.astype(np.int64, copy=False)
            symbols = np.unique(p)
            if len(symbols) > 255:
                raise UnsupportedSearch('subsequences with more than 255 distinct symbols are not supported')

            def code(a):
                if len(a) == 0:
                    return b''
                pos = np.minimum(np.searchsorted(symbols, a), len(symbols) - 1)
                return np.where(symbols[pos] == a, pos + 1, 0).astype(np.uint8).tobytes()
            return code(p), code(t)
    except (ValueError, TypeError, OverflowError):
        pass                                             # ragged / mixed items: the general form below
    table = {}
    for item in subsequence:
        if item not in table:
            if len(table) == 255:
                raise UnsupportedSearch('subsequences with more than 255 distinct symbols are not supported')
            table[item] = len(table) + 1
    from itertools import repeat
    return bytes(map(table.__getitem__, subsequence)), bytes(map(table.get, sequence, repeat(0)))


def encode_pair(subsequence, sequence):
    """-> (pattern_bytes_like, sequence_bytes_like, byteslike: bool)."""
    subsequence, bio_p = _unwrap_bio(subsequence)
    sequence, bio_t = _unwrap_bio(sequence)
    if bio_p or bio_t:                               # a Seq on either side: compare letters (str patterns as latin-1)
        if isinstance(subsequence, str):
            subsequence = subsequence.encode('latin-1')
        if isinstance(sequence, str):
            sequence = sequence.encode('latin-1')
        return subsequence, sequence, False
    sub_b, seq_b = is_byteslike(subsequence), is_byteslike(sequence)
    if sub_b and seq_b:
        return subsequence, sequence, True
    if isinstance(subsequence, str) and isinstance(sequence, str):
        try:
            return subsequence.encode('latin-1'), sequence.encode('latin-1'), False
        except UnicodeEncodeError:
            p, t = _remap_str(subsequence, sequence)
            return p, t, False
    if isinstance(sequence, (list, tuple)):
        p, t = _remap_items(subsequence, sequence)
        return p, t, False
    raise TypeError('unsupported combination of subsequence / sequence types: %s / %s'
                    % (type(subsequence).__name__, type(sequence).__name__))


class DeviceSequence(object):
    """A sequence made resident in HBM once; pass it as ``sequence`` to find_near_matches().

    Only bytes-like and latin-1 ``str`` sequences can be made resident ahead of the query (symbol
    remapping depends on the subsequence).  ``len()``, slicing and ``matched`` come from the
    original object, which is kept alive.
    """

    def __init__(self, sequence, engine=None):
        self.original = sequence
        self.engine = engine or _native.default_engine()
        if is_byteslike(sequence):
            data = sequence
            self.byteslike = True
        elif isinstance(sequence, str):
            data = sequence.encode('latin-1')      # UnicodeEncodeError -> not residency-capable
            self.byteslike = False
        else:
            raise TypeError('only bytes-like or latin-1 str sequences can be made resident')
        self.handle = self.engine.upload(data)

    def __len__(self):
        return len(self.original)

    def __getitem__(self, item):
        return self.original[item]

    def release(self):
        self.handle.release()


def resident(sequence, engine=None):
    """Upload ``sequence`` to HBM and return a handle usable wherever a sequence is expected."""
    return DeviceSequence(sequence, engine)


class ResidencyCache(object):
    """Transparent residency for the reference's call form, ``find_near_matches(subsequence, sequence)`` with a plain
    sequence (__init__.py:35-57): the last few uploaded IMMUTABLE sequences stay in HBM, so a second query against the
    same object uploads nothing ("device-buffer cache so a sequence can stay resident across
    queries").  Only exact ``bytes`` and ``str`` objects qualify — what they hold cannot change, and the cache keeps a
    strong reference, so an ``id()`` cannot be handed to another object while its entry lives; ``bytearray``,
    ``memoryview``, numpy arrays, lists ... are uploaded on every call as before.  A ``str`` entry holds its latin-1
    bytes (symbol-remapped text depends on the subsequence and is never cached).

    The reference holds nothing once it has returned (__init__.py:35-57), and neither does the cache for long: every
    call first drops the entries whose object only the cache still references (the caller has let go — such an entry can
    never be hit again), so a dropped sequence gives its host bytes and its HBM back at the next search, whatever the
    budget.  Beyond that, least recently used entries leave BEFORE a new sequence is uploaded, until it fits the byte
    budget and the entry count; an upload that fails all the same empties the cache and is tried once more; an entry that
    a search is still using is released by its last user.
    Budget: FUZZYSEARCH_HIP_RESIDENT_CACHE (bytes, K / M / G suffixes; 0 switches the cache off); by default a quarter of the
    device memory that is free when the cache is first used, at most 8G.  ``fuzzysearch_amd.cache_info()`` /
    ``cache_clear()`` are the public handles."""

    MIN_BYTES = 1 << 16              # below this an upload costs less than it is worth tracking
    MAX_ENTRIES = 16
    DEFAULT_MAX = 8 << 30
    DEFAULT_FRACTION = 0.25          # of the free device memory at first use

    class _Entry(object):
        __slots__ = ('obj', 'handle', 'nbytes', 'users', 'evicted', 'engine')

    def __init__(self, budget=None):
        import collections
        import threading
        self._lock = threading.Lock()
        self._entries = collections.OrderedDict()        # id(obj) -> _Entry, least recently used first
        self._bytes = 0
        self._budget = self._env_budget() if budget is None else int(budget)     # None: derived from the device at first use
        self.hits = self.misses = self.evictions = self.orphans = self.retries = 0
        self._tls = threading.local()

    @staticmethod
    def _env_budget():
        import os
        raw = os.environ.get('FUZZYSEARCH_HIP_RESIDENT_CACHE', '').strip().upper()
        if not raw:
            return None
        mult = 1
        if raw[-1] in 'KMG':
            mult = 1 << (10 * (1 + 'KMG'.index(raw[-1])))
            raw = raw[:-1]
        try:
            return max(0, int(float(raw) * mult))
        except ValueError:
            return None

    @property
    def budget(self):
        """Bytes the cache may hold.  Not set by the environment: DEFAULT_FRACTION of the free device memory, found when the
        cache first needs the number (it takes a device), capped at DEFAULT_MAX."""
        if self._budget is None:
            try:
                free, _total = _native.default_engine().mem_info()
                self._budget = min(self.DEFAULT_MAX, int(free * self.DEFAULT_FRACTION))
            except Exception:                            # no device yet / no such entry point: decide again next time
                return self.DEFAULT_MAX
        return self._budget

    @budget.setter
    def budget(self, value):
        self._budget = None if value is None else int(value)

    @staticmethod
    def cacheable(sequence):
        return type(sequence) is bytes or type(sequence) is str

    def bypassed(self):
        """Inside a `with cache.bypass():` block of this thread (the per-chunk file searches: every chunk is a new object that
        is searched once)."""
        return getattr(self._tls, 'off', 0) > 0

    def bypass(self):
        cache = self

        class _Bypass(object):
            def __enter__(self_inner):
                cache._tls.off = getattr(cache._tls, 'off', 0) + 1

            def __exit__(self_inner, *exc):
                cache._tls.off -= 1
        return _Bypass()

    def sweep(self):
        """Drop the entries whose objects nobody but the cache references any more; -> how many."""
        dead = []
        with self._lock:
            if self._entries:
                self._sweep(dead)
        for h in dead:
            h.release()
        return len(dead)

    def _sweep(self, dead, keep=None):
        """Under the lock: drop every entry whose object is referenced by nothing but its entry (the attribute and
        getrefcount's own argument make 2)."""
        import sys
        for key in [k for k, e in self._entries.items() if e.obj is not keep and e.users == 0 and sys.getrefcount(e.obj) <= 2]:
            self._drop(key, dead)
            self.orphans += 1

    def acquire(self, engine, sequence, make_data):
        """-> (handle, entry or None).  `make_data()`: the bytes to upload for `sequence` (itself, or its latin-1
        encoding: one byte per item either way), only called when the sequence is not resident.
        entry None: not cached, the caller owns the handle; otherwise the caller calls done(entry) after its search."""
        n = len(sequence)
        dead = []
        try:
            if self.bypassed() or not self.cacheable(sequence) or n < self.MIN_BYTES:
                with self._lock:
                    self._sweep(dead)
                return self._upload(engine, make_data), None
            budget = self.budget
            key = id(sequence)
            with self._lock:
                self._sweep(dead, keep=sequence)
                e = self._entries.get(key)
                if e is not None and e.obj is sequence and e.engine is engine and budget > 0:
                    self._entries.move_to_end(key)
                    e.users += 1
                    self.hits += 1
                    return e.handle, e
                if budget <= 0 or n > budget:
                    cached = False
                else:
                    cached = True
                    # room first: the new sequence is uploaded into memory the cache has already given back
                    while self._entries and (self._bytes + n > budget or len(self._entries) + 1 > self.MAX_ENTRIES):
                        self._drop(next(iter(self._entries)), dead)
            for h in dead:
                h.release()
            del dead[:]
            handle = self._upload(engine, make_data)         # outside the lock: tens of milliseconds per GiB
            if not cached:
                return handle, None
            e = self._Entry()
            e.obj, e.handle, e.nbytes, e.users, e.evicted, e.engine = sequence, handle, n, 1, False, engine
            with self._lock:
                self.misses += 1
                if key in self._entries:                     # another thread uploaded the same object meanwhile: keep the newer
                    self._drop(key, dead)
                self._entries[key] = e
                self._bytes += n
            return handle, e
        finally:
            for h in dead:
                h.release()

    def _upload(self, engine, make_data):
        """One upload; if the device has no room for it, everything the cache holds goes and it is tried once more."""
        try:
            return engine.upload(make_data())
        except (_native.HipEngineError, MemoryError):
            if not self._entries:
                raise
            self.retries += 1
            self.clear()
            return engine.upload(make_data())

    def _drop(self, key, dead):
        e = self._entries.pop(key)
        self._bytes -= e.nbytes
        self.evictions += 1
        e.evicted = True
        e.obj = None
        if e.users == 0:
            dead.append(e.handle)

    def done(self, entry):
        release = False
        with self._lock:
            entry.users -= 1
            release = entry.evicted and entry.users == 0
        if release:
            entry.handle.release()

    def clear(self):
        dead = []
        with self._lock:
            for key in list(self._entries):
                self._drop(key, dead)
        for h in dead:
            h.release()

    def info(self):
        with self._lock:
            return {'entries': len(self._entries), 'bytes': self._bytes, 'budget': self._budget, 'hits': self.hits,
                    'misses': self.misses, 'evictions': self.evictions, 'orphans_dropped': self.orphans,
                    'upload_retries': self.retries}


_cache = ResidencyCache()


def residency_cache():
    """The process-wide cache behind find_near_matches(subsequence, <bytes or str>): .info(), .clear(), .budget."""
    return _cache


def cache_info():
    """What the residency cache holds: {'entries', 'bytes', 'budget' (None: not derived from the device yet), 'hits',
    'misses', 'evictions', 'orphans_dropped', 'upload_retries'}."""
    return _cache.info()


def cache_clear():
    """Release every cached sequence (host reference and HBM) now."""
    _cache.clear()


class _Prepared(object):
    """(engine, resident handle, pattern bytes, original sequence, byteslike flag) for one query."""
    __slots__ = ('engine', 'handle', 'pattern', 'original', 'byteslike', 'owned', 'entry')

    def release(self):
        if self.entry is not None:
            _cache.done(self.entry)
            self.entry = None
        elif self.owned:
            self.handle.release()


def prepare(subsequence, sequence):
    pr = _Prepared()
    pr.entry = None
    if isinstance(sequence, DeviceSequence):
        subsequence, was_bio = _unwrap_bio(subsequence)
        if was_bio and not sequence.byteslike:
            subsequence = subsequence.decode('latin-1')
        pr.engine = sequence.engine
        pr.handle = sequence.handle
        pr.original = sequence.original
        pr.byteslike = sequence.byteslike
        pr.owned = False
        if sequence.byteslike:
            if not is_byteslike(subsequence):
                raise TypeError('a bytes-like subsequence is required for a bytes-like sequence')
            pr.pattern = subsequence
        else:
            if not isinstance(subsequence, str):
                raise TypeError('a str subsequence is required for a str sequence')
            try:
                pr.pattern = subsequence.encode('latin-1')
            except UnicodeEncodeError:
                # a symbol the resident latin-1 text cannot contain: it can never match anything.
                # 0xFF..0x00 trick is unsafe, so fall back to a per-query upload with remapping.
                p, t, _ = encode_pair(subsequence, sequence.original)
                pr.engine = sequence.engine
                pr.handle = pr.engine.upload(t)
                pr.pattern = p
                pr.owned = True
        return pr
    pr.engine = _native.default_engine()
    pr.original = sequence
    pr.owned = True
    _cache.sweep()                                       # sequences their owners have dropped give HBM and host bytes back now
    # the reference's call form on an immutable sequence that is already resident: nothing to encode or upload
    if len(sequence) >= ResidencyCache.MIN_BYTES and not _cache.bypassed() and _cache.budget > 0:
        if type(sequence) is bytes and is_byteslike(subsequence):
            pr.pattern, pr.byteslike = subsequence, True
            pr.handle, pr.entry = _cache.acquire(pr.engine, sequence, lambda: sequence)
            return pr
        if type(sequence) is str and type(subsequence) is str:
            try:
                pr.pattern, pr.byteslike = subsequence.encode('latin-1'), False
                pr.handle, pr.entry = _cache.acquire(pr.engine, sequence, lambda: sequence.encode('latin-1'))
                return pr
            except UnicodeEncodeError:
                pass                                     # wide code points: symbol remapping below, never cached
    p, t, byteslike = encode_pair(subsequence, sequence)
    pr.handle = pr.engine.upload(t)
    pr.pattern = p
    pr.byteslike = byteslike
    return pr
