"""ctypes binding of libfzhip.so (include/fzhip.h).  No PyTorch, no CPU fallback.

The library is built in-tree by ``fuzzysearch_amd/build.py`` (hipcc, gfx950).  If it is missing or
there is no MI355X, every search raises — this package never computes matches on the CPU.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# FUZZYSEARCH_HIP_LIB: another build of the same C-ABI (A/B benchmarking of kernel versions)
LIB_PATH = os.environ.get("FUZZYSEARCH_HIP_LIB") or os.path.join(_HERE, "libfzhip.so")

FZ_OK, FZ_EINVAL, FZ_ENOMEM, FZ_EDEVICE, FZ_EUNSUPPORTED, FZ_EHALO, FZ_ETIMEOUT = 0, -1, -2, -3, -4, -5, -6
UINT64_MAX = (1 << 64) - 1

# every symbol include/fzhip.h declares (tests check the library exports exactly these)
EXPORTED_SYMBOLS = (
    "fz_abi_version", "fz_last_error", "fz_device_count", "fz_create", "fz_destroy",
    "fz_seq_upload", "fz_seq_upload_shard", "fz_seq_new", "fz_seq_add_shard", "fz_seq_len", "fz_seq_release",
    "fz_search_exact", "fz_lev_ngrams", "fz_lev_ngrams_begin", "fz_lev_ngrams_end", "fz_subs_ngrams", "fz_generic_ngrams",
    "fz_subs_ngrams_begin", "fz_generic_ngrams_begin", "fz_search_end",
    "fz_lev_lp", "fz_subs_lp", "fz_generic_lp", "fz_subs_ngrams_any", "fz_subs_lp_any", "fz_generic_ngrams_any", "fz_generic_ngrams_consolidated",
    "fz_lev_ngrams_consolidated", "fz_subs_ngrams_best",
    "fz_stream_open", "fz_stream_buffer", "fz_stream_submit", "fz_stream_read_fd", "fz_stream_finish", "fz_stream_close",
    "fz_consolidate", "fz_group_best", "fz_merge_ranks", "fz_wire_pack", "fz_wire_merge", "fz_debug_launch_plan", "fz_debug_order_records", "fz_debug_order_records_bounded", "fz_debug_order_segments", "fz_stats", "fz_mem_info", "fz_set_timing", "fz_set_streams", "fz_device_ms", "fz_free",
    "fz_comm_unique_id", "fz_comm_init_rank", "fz_comm_init_all", "fz_comm_info", "fz_comm_set_collective",
    "fz_comm_allgather", "fz_comm_max_f64", "fz_comm_barrier", "fz_comm_destroy", "fz_comm_gather_ms", "fz_comm_backend", "fz_debug_reload_switches", "fz_debug_gather_merge", "fz_debug_scan_regions",
)


class FzMatch(ctypes.Structure):
    _fields_ = [("start", ctypes.c_int64), ("end", ctypes.c_int64),
                ("dist", ctypes.c_int32), ("block", ctypes.c_int32)]


class FzStats(ctypes.Structure):
    _fields_ = [("bytes_scanned", ctypes.c_uint64), ("ngram_hits", ctypes.c_uint64),
                ("raw_matches", ctypes.c_uint64), ("filter_ms", ctypes.c_double),
                ("verify_ms", ctypes.c_double), ("device_ms", ctypes.c_double),
                ("filter_launches", ctypes.c_uint32), ("n_devices", ctypes.c_uint32),
                ("verify_form", ctypes.c_uint32), ("reserved_", ctypes.c_uint32)]


# FzStats.verify_form (include/fzhip.h: FZ_FORM_*)
FORM_NONE, FORM_FUSED_BAND, FORM_FUSED_CELLS, FORM_FUSED_BITS1, FORM_FUSED_BITS2, FORM_KERNEL, FORM_FUSED_BITS32 = range(7)


class HipEngineError(RuntimeError):
    """libfzhip.so missing / no usable gfx950 device / HIP runtime failure."""


class CollectiveTimeout(HipEngineError):
    """A collective of the context's communicator did not complete within FZ_COMM_TIMEOUT_MS (a rank never arrived)."""


_lib = None
_lib_lock = threading.Lock()


def load_library():
    """Load libfzhip.so and declare the C-ABI.  Raises HipEngineError if it is not built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipEngineError(
                "fuzzysearch_amd: %s is missing. Build it with `python -m fuzzysearch_amd.build` "
                "(hipcc, --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        vp, u8p = ctypes.c_void_p, ctypes.c_void_p
        u32, u64, ci = ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
        mpp = ctypes.POINTER(ctypes.POINTER(FzMatch))
        u64p = ctypes.POINTER(u64)
        L.fz_abi_version.restype = ci
        L.fz_abi_version.argtypes = []
        L.fz_last_error.restype = ctypes.c_char_p
        L.fz_last_error.argtypes = []
        L.fz_device_count.restype = ci
        L.fz_device_count.argtypes = [ctypes.POINTER(ci)]
        L.fz_create.restype = ci
        L.fz_create.argtypes = [ctypes.POINTER(ci), ci, ctypes.POINTER(vp)]
        L.fz_destroy.restype = None
        L.fz_destroy.argtypes = [vp]
        L.fz_seq_upload.restype = ci
        L.fz_seq_upload.argtypes = [vp, u8p, u64, ctypes.POINTER(vp)]
        L.fz_seq_upload_shard.restype = ci
        L.fz_seq_upload_shard.argtypes = [vp, u8p, u64, u64, u64, u64, u64, ctypes.POINTER(vp)]
        L.fz_seq_new.restype = ci
        L.fz_seq_new.argtypes = [vp, u64, ctypes.POINTER(vp)]
        L.fz_seq_add_shard.restype = ci
        L.fz_seq_add_shard.argtypes = [vp, ci, u8p, u64, u64, u64, u64]
        L.fz_device_ms.restype = ci
        L.fz_device_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ci]
        L.fz_comm_unique_id.restype = ci
        L.fz_comm_unique_id.argtypes = [vp, u64]
        L.fz_comm_init_rank.restype = ci
        L.fz_comm_init_rank.argtypes = [vp, vp, ci, ci]
        L.fz_comm_init_all.restype = ci
        L.fz_comm_init_all.argtypes = [vp]
        L.fz_comm_info.restype = ci
        L.fz_comm_info.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.fz_comm_set_collective.restype = ci
        L.fz_comm_set_collective.argtypes = [vp, ci]
        L.fz_comm_allgather.restype = ci
        L.fz_comm_allgather.argtypes = [vp, vp, u64, vp]
        L.fz_comm_max_f64.restype = ci
        L.fz_comm_max_f64.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.fz_comm_barrier.restype = ci
        L.fz_comm_barrier.argtypes = [vp]
        L.fz_comm_destroy.restype = None
        L.fz_comm_destroy.argtypes = [vp]
        L.fz_debug_reload_switches.restype = None
        L.fz_debug_reload_switches.argtypes = []
        L.fz_comm_backend.restype = ci
        L.fz_comm_backend.argtypes = []
        L.fz_comm_gather_ms.restype = ci
        L.fz_comm_gather_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.fz_debug_scan_regions.restype = ci
        L.fz_debug_scan_regions.argtypes = [u64, u64, u32, ci, ctypes.c_double, ci, ctypes.POINTER(u32), ctypes.c_void_p]
        L.fz_debug_gather_merge.restype = ci
        L.fz_debug_gather_merge.argtypes = [ctypes.c_void_p, u32, u64, ctypes.c_void_p, u32, mpp, u64p, u64p]
        L.fz_seq_len.restype = u64
        L.fz_seq_len.argtypes = [vp]
        L.fz_seq_release.restype = None
        L.fz_seq_release.argtypes = [vp]
        L.fz_search_exact.restype = ci
        L.fz_search_exact.argtypes = [vp, vp, u8p, u32, u64, u64, ctypes.POINTER(u64p), u64p]
        L.fz_lev_ngrams.restype = ci
        L.fz_lev_ngrams.argtypes = [vp, vp, u8p, u32, u32, mpp, u64p]
        L.fz_lev_ngrams_begin.restype = ci
        L.fz_lev_ngrams_begin.argtypes = [vp, vp, u8p, u32, u32]
        L.fz_lev_ngrams_end.restype = ci
        L.fz_lev_ngrams_end.argtypes = [vp, mpp, u64p]
        L.fz_search_end.restype = ci
        L.fz_search_end.argtypes = [vp, mpp, u64p]
        L.fz_subs_ngrams_begin.restype = ci
        L.fz_subs_ngrams_begin.argtypes = [vp, vp, u8p, u32, u32]
        L.fz_generic_ngrams_begin.restype = ci
        L.fz_generic_ngrams_begin.argtypes = [vp, vp, u8p, u32, u32, u32, u32, u32, ci]
        L.fz_subs_ngrams.restype = ci
        L.fz_subs_ngrams.argtypes = [vp, vp, u8p, u32, u32, mpp, u64p]
        L.fz_generic_ngrams.restype = ci
        L.fz_generic_ngrams.argtypes = [vp, vp, u8p, u32, u32, u32, u32, u32, mpp, u64p]
        L.fz_lev_lp.restype = ci
        L.fz_lev_lp.argtypes = [vp, vp, u8p, u32, u32, mpp, u64p]
        L.fz_subs_lp.restype = ci
        L.fz_subs_lp.argtypes = [vp, vp, u8p, u32, u32, mpp, u64p]
        L.fz_generic_lp.restype = ci
        L.fz_generic_lp.argtypes = [vp, vp, u8p, u32, u32, u32, u32, u32, mpp, u64p]
        L.fz_lev_ngrams_consolidated.restype = ci
        L.fz_lev_ngrams_consolidated.argtypes = [vp, vp, u8p, u32, u32, mpp, u64p]
        L.fz_subs_ngrams_best.restype = ci
        L.fz_subs_ngrams_best.argtypes = [vp, vp, u8p, u32, u32, mpp, u64p]
        L.fz_generic_ngrams_consolidated.restype = ci
        L.fz_generic_ngrams_consolidated.argtypes = [vp, vp, u8p, u32, u32, u32, u32, u32, mpp, u64p]
        L.fz_subs_ngrams_any.restype = ci
        L.fz_subs_ngrams_any.argtypes = [vp, vp, u8p, u32, u32, ctypes.POINTER(ci)]
        L.fz_subs_lp_any.restype = ci
        L.fz_subs_lp_any.argtypes = [vp, vp, u8p, u32, u32, ctypes.POINTER(ci)]
        L.fz_generic_ngrams_any.restype = ci
        L.fz_generic_ngrams_any.argtypes = [vp, vp, u8p, u32, u32, u32, u32, u32, ctypes.POINTER(ci)]
        L.fz_stream_open.restype = ci
        L.fz_stream_open.argtypes = [vp, u32, u8p, u32, u32, u32, u32, u32, u64, u32, u32, u64, ctypes.POINTER(vp)]
        L.fz_stream_buffer.restype = ci
        L.fz_stream_buffer.argtypes = [vp, ctypes.POINTER(vp), u64p]
        L.fz_stream_submit.restype = ci
        L.fz_stream_submit.argtypes = [vp, u64, ci]
        L.fz_stream_read_fd.restype = ci
        L.fz_stream_read_fd.argtypes = [vp, ci, ctypes.c_int64, ci, u64p]
        L.fz_stream_finish.restype = ci
        L.fz_stream_finish.argtypes = [vp, mpp, ctypes.POINTER(ctypes.POINTER(u32)), u64p]
        L.fz_stream_close.restype = None
        L.fz_stream_close.argtypes = [vp]
        L.fz_consolidate.restype = ci
        L.fz_consolidate.argtypes = [ctypes.POINTER(FzMatch), u64, mpp, u64p]
        L.fz_group_best.restype = ci
        L.fz_group_best.argtypes = [ctypes.POINTER(FzMatch), u64, mpp, u64p]
        L.fz_wire_pack.restype = ci
        L.fz_wire_pack.argtypes = [ctypes.c_void_p, u64, u64, ctypes.c_void_p]
        L.fz_wire_merge.restype = ci
        L.fz_wire_merge.argtypes = [ctypes.c_void_p, u32, u64, u64, ctypes.c_void_p, u64, u64p, u64p]
        L.fz_debug_launch_plan.restype = ci
        L.fz_debug_launch_plan.argtypes = [u8p, u32, u32, ctypes.POINTER(u32), u32, ctypes.POINTER(u32)]
        L.fz_debug_order_records.restype = ci
        L.fz_debug_order_records.argtypes = [ctypes.c_void_p, u64, u32, mpp, u64p]
        L.fz_debug_order_records_bounded.restype = ci
        L.fz_debug_order_records_bounded.argtypes = [ctypes.c_void_p, u64, u32, u64, u32, mpp, u64p]
        L.fz_debug_order_segments.restype = ci
        L.fz_debug_order_segments.argtypes = [ctypes.c_void_p, ctypes.c_void_p, u32, u32, mpp, u64p]
        L.fz_merge_ranks.restype = ci
        L.fz_merge_ranks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, u32, u32, ctypes.c_void_p]
        L.fz_stats.restype = ci
        L.fz_stats.argtypes = [vp, ctypes.POINTER(FzStats)]
        if hasattr(L, "fz_mem_info"):                       # (absent from builds of earlier rounds named by FUZZYSEARCH_HIP_LIB for an A/B)
            L.fz_mem_info.restype = ci
            L.fz_mem_info.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        L.fz_set_timing.restype = ci
        L.fz_set_timing.argtypes = [vp, ci]
        L.fz_set_streams.restype = ci
        L.fz_set_streams.argtypes = [vp, ci]
        L.fz_free.restype = None
        L.fz_free.argtypes = [vp]
        if L.fz_abi_version() != 1:
            raise HipEngineError("libfzhip.so ABI version mismatch")
        _lib = L
        return L


class UnsupportedSearch(NotImplementedError):
    """The search is outside what the GPU engine supports: a subsequence of more than 65 535 items, a Levenshtein
    budget above 1 023, a generic search (separate limits) or a linear-programming route with max_l_dist above 255,
    more than 255 distinct symbols in a subsequence that is neither bytes nor latin-1 text, automaton candidate sets
    beyond 2^18 entries.  Nothing was searched and there is no CPU fallback: a caller that needs such a search
    catches this and routes it to the reference implementation."""


def _raise(rc):
    msg = (load_library().fz_last_error() or b"").decode("utf-8", "replace")
    if rc == FZ_EINVAL:
        raise ValueError(msg)
    if rc == FZ_ENOMEM:
        raise MemoryError(msg)
    if rc == FZ_EUNSUPPORTED:
        raise UnsupportedSearch(msg)
    if rc == FZ_ETIMEOUT:
        raise CollectiveTimeout("libfzhip error %d: %s" % (rc, msg))
    raise HipEngineError("libfzhip error %d: %s" % (rc, msg))


def _check(rc):
    if rc != FZ_OK:
        _raise(rc)


def _buffer_address(data):
    """-> (address, nbytes, keepalive) of a C-contiguous 1-byte-item buffer, zero-copy where the
    buffer protocol allows (bytes, bytearray, memoryview, numpy uint8 ...).  Same acceptance rule
    as the reference's is_simple_buffer (_c_ext_base.h:27-34)."""
    mv = memoryview(data)
    if mv.itemsize != 1 or mv.ndim != 1 or not mv.c_contiguous:
        raise TypeError("only contiguous sequences of single-byte values are supported")
    n = mv.nbytes
    if n == 0:
        return None, 0, mv
    if mv.readonly:
        # ctypes cannot wrap a read-only buffer without a copy unless it is a bytes object
        obj = mv.obj
        if isinstance(obj, bytes):
            return ctypes.cast(ctypes.c_char_p(obj), ctypes.c_void_p).value + _mv_offset(mv, obj), n, (mv, obj)
        import numpy as np
        arr = np.frombuffer(mv, dtype=np.uint8)
        return arr.ctypes.data, n, (mv, arr)
    c = (ctypes.c_char * n).from_buffer(mv)
    return ctypes.addressof(c), n, (mv, c)


def _mv_offset(mv, obj):
    # offset of a memoryview slice into its bytes object (memoryview(b)[a:b])
    if len(mv) == len(obj):
        return 0
    import numpy as np
    base = np.frombuffer(obj, dtype=np.uint8).ctypes.data
    return np.frombuffer(mv, dtype=np.uint8).ctypes.data - base


def matches_to_array(raw):
    """(start, end, dist[, block]) rows, or an fz_match structured array -> (address-able buffer, n)."""
    import numpy as np
    if isinstance(raw, np.ndarray) and raw.dtype == _match_dtype():
        arr = np.ascontiguousarray(raw)
        return arr, len(arr)
    n = len(raw)
    arr = np.empty(n, dtype=_match_dtype())
    if n:
        rows = np.asarray([tuple(r[:3]) for r in raw], dtype=np.int64).reshape(n, 3)
        arr["start"], arr["end"], arr["dist"] = rows[:, 0], rows[:, 1], rows[:, 2]
        arr["block"] = [r[3] if len(r) > 3 else -1 for r in raw]
    return arr, n


_MATCH_DTYPE = None


def _match_dtype():
    global _MATCH_DTYPE
    if _MATCH_DTYPE is None:
        import numpy as np
        _MATCH_DTYPE = np.dtype([("start", "<i8"), ("end", "<i8"), ("dist", "<i4"), ("block", "<i4")])
    return _MATCH_DTYPE


class _ResultBuffer(object):
    """Owner of a C-ABI result buffer that a numpy array views in place; fz_free when the last view is gone."""
    __slots__ = ('_lib', '_addr')

    def __init__(self, lib, addr):
        self._lib, self._addr = lib, addr

    def __del__(self):
        lib, addr = self._lib, self._addr
        self._addr = None
        if addr and lib is not None:
            try:
                lib.fz_free(ctypes.c_void_p(addr))
            except Exception:                                   # interpreter shutdown
                pass


def _take_matches_array(L, ptr, n):
    """-> numpy structured array (start, end, dist, block).  Small results: one memcpy, then the C buffer is
    freed.  Large ones (>= 256 KiB: e.g. the 2.1e5 rows of a generic search over 1 GiB of text) are viewed in
    place — no second 5 MB buffer to fault in and fill — and handed back to the library (which recycles such
    buffers) when the last view dies."""
    import numpy as np
    nbytes = n * ctypes.sizeof(FzMatch)
    if nbytes >= (256 << 10):
        addr = ctypes.cast(ptr, ctypes.c_void_p).value
        raw = (ctypes.c_char * nbytes).from_address(addr)
        raw._owner = _ResultBuffer(L, addr)                     # lives as long as the ctypes view numpy holds on to
        return np.frombuffer(raw, dtype=_match_dtype(), count=n)
    arr = np.empty(n, dtype=_match_dtype())
    if n:
        ctypes.memmove(arr.ctypes.data, ptr, nbytes)
    L.fz_free(ptr)
    return arr


def _take_matches(L, ptr, n):
    return _take_matches_array(L, ptr, n).tolist()


def _array_call(fn, raw):
    L = load_library()
    arr, n = matches_to_array(raw)
    ptr = ctypes.POINTER(FzMatch)()
    cnt = ctypes.c_uint64(0)
    _check(fn(L)(ctypes.cast(arr.ctypes.data, ctypes.POINTER(FzMatch)), n, ctypes.byref(ptr), ctypes.byref(cnt)))
    return _take_matches_array(L, ptr, cnt.value)


def consolidate_array(raw):
    """consolidate_overlapping_matches (common.py:185-189) on an fz_match array (or rows) -> fz_match array;
    no per-record Python work."""
    return _array_call(lambda L: L.fz_consolidate, raw)


def group_best_array(raw):
    """[get_best_match_in_group(g) for g in group_matches(ms)] in group-list order
    (substitutions_only.py:279-282) on an fz_match array (or rows) -> fz_match array."""
    return _array_call(lambda L: L.fz_group_best, raw)


def consolidate(raw):
    """consolidate_overlapping_matches on (start, end, dist[, block]) tuples -> list of tuples."""
    return consolidate_array(raw).tolist()


def group_best(raw):
    return group_best_array(raw).tolist()


class OwnedRows(object):
    """A C-ABI result buffer (fz_match rows) that has not been copied anywhere: address, n, and fz_free on release."""
    __slots__ = ('_lib', '_ptr', 'n')

    def __init__(self, lib, ptr, n):
        self._lib, self._ptr, self.n = lib, ptr, n

    @property
    def address(self):
        return ctypes.cast(self._ptr, ctypes.c_void_p).value or 0

    def to_array(self):
        ptr, self._ptr = self._ptr, None
        return _take_matches_array(self._lib, ptr, self.n)

    def release(self):
        ptr, self._ptr = self._ptr, None
        if ptr is not None:
            self._lib.fz_free(ptr)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class FileStream(object):
    """fz_stream: the chunks of a file searched in batches (include/fzhip.h)."""

    def __init__(self, engine, mode, pattern, limits, k, seg_stride, seg_pre, seg_post, batch_bytes):
        self.engine = engine
        self._lib = engine._lib
        self._h = None
        paddr, m, keep = _buffer_address(pattern)
        h = ctypes.c_void_p()
        ms, mi, md = limits
        with engine._lock:
            _check(self._lib.fz_stream_open(engine._h, mode, paddr, m, ms, mi, md, k, seg_stride, seg_pre, seg_post,
                                            batch_bytes, ctypes.byref(h)))
        del keep
        self._h = h

    def buffer(self):
        """-> writable memoryview of the free part of the current pinned staging buffer."""
        ptr = ctypes.c_void_p()
        cap = ctypes.c_uint64(0)
        _check(self._lib.fz_stream_buffer(self._h, ctypes.byref(ptr), ctypes.byref(cap)))
        if cap.value == 0:
            return memoryview(bytearray(0))
        return memoryview((ctypes.c_char * cap.value).from_address(ptr.value)).cast('B')

    def submit(self, nbytes, last=False):
        with self.engine._lock:
            _check(self._lib.fz_stream_submit(self._h, nbytes, 1 if last else 0))

    def read_fd(self, fd, offset, threads=0):
        total = ctypes.c_uint64(0)
        with self.engine._lock:
            _check(self._lib.fz_stream_read_fd(self._h, fd, offset, threads, ctypes.byref(total)))
        return total.value

    def finish(self):
        """-> (fz_match structured array in the reference's order, uint32 chunk number per record)."""
        import numpy as np
        ptr = ctypes.POINTER(FzMatch)()
        seg = ctypes.POINTER(ctypes.c_uint32)()
        cnt = ctypes.c_uint64(0)
        with self.engine._lock:
            _check(self._lib.fz_stream_finish(self._h, ctypes.byref(ptr), ctypes.byref(seg), ctypes.byref(cnt)))
        n = cnt.value
        segs = np.empty(n, dtype=np.uint32)
        if n:
            ctypes.memmove(segs.ctypes.data, seg, 4 * n)
        self._lib.fz_free(seg)
        return _take_matches_array(self._lib, ptr, n), segs

    def close(self):
        with self.engine._lock:                    # the engine may be closed (fz_destroy) by another thread: test under the lock
            if self._h is not None and self.engine._h is not None:
                self._lib.fz_stream_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResidentSequence(object):
    """A sequence resident in HBM (fz_seq).  Keeps the engine alive; release() or GC frees it."""

    def __init__(self, engine, handle, nbytes):
        self.engine = engine
        self._h = handle
        self.nbytes = nbytes

    def __len__(self):
        return self.nbytes

    def release(self):
        with self.engine._lock:                    # fz_destroy frees live sequences: test the engine under its lock
            if self._h is not None and self.engine._h is not None:
                self.engine._lib.fz_seq_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Engine(object):
    """One fz_ctx: HIP streams, hit/record buffers and the resident sequences of one host thread."""

    def __init__(self, devices=None):
        self._lib = load_library()
        self._h = None
        h = ctypes.c_void_p()
        if devices:
            ids = (ctypes.c_int * len(devices))(*devices)
            rc = self._lib.fz_create(ids, len(devices), ctypes.byref(h))
        else:
            rc = self._lib.fz_create(None, 0, ctypes.byref(h))
        _check(rc)
        self._h = h
        self.devices = list(devices) if devices else [0]
        # a fz_ctx is not internally locked and ctypes drops the GIL during calls: serialise per engine
        # re-entrant: the cyclic GC may finalize a ResidentSequence (-> release()) on a thread that already holds it
        self._lock = threading.RLock()
        self._st = FzStats()
        self._st_ref = ctypes.byref(self._st)

    def close(self):
        """Destroy the context.  Sequences that are still resident are freed with it (fz_destroy); their
        ResidentSequence handles become inert (release() checks the engine)."""
        if self._h is not None:
            with self._lock:
                self._lib.fz_destroy(self._h)
                self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- residency ---------------------------------------------------------------------------
    def upload(self, data):
        addr, n, keep = _buffer_address(data)
        h = ctypes.c_void_p()
        with self._lock:
            _check(self._lib.fz_seq_upload(self._h, addr, n, ctypes.byref(h)))
        del keep
        return ResidentSequence(self, h, n)

    def upload_shard(self, data, buf_global_off, own_lo, own_hi, global_n):
        addr, n, keep = _buffer_address(data)
        h = ctypes.c_void_p()
        with self._lock:
            _check(self._lib.fz_seq_upload_shard(self._h, addr, n, buf_global_off, own_lo, own_hi,
                                                  global_n, ctypes.byref(h)))
        del keep
        return ResidentSequence(self, h, global_n)

    def new_sequence(self, global_n):
        """An empty sharded sequence of global_n bytes; fill it with add_shard() (one shard per device of the engine)."""
        h = ctypes.c_void_p()
        with self._lock:
            _check(self._lib.fz_seq_new(self._h, global_n, ctypes.byref(h)))
        return ResidentSequence(self, h, global_n)

    def add_shard(self, seq, dev_index, data, buf_global_off, own_lo, own_hi):
        """Upload bytes [buf_global_off, +len(data)) to device number dev_index of this engine; it owns the n-gram
        hits in [own_lo, own_hi) and needs (m + k) halo bytes on both sides (fz_seq_add_shard)."""
        addr, n, keep = _buffer_address(data)
        with self._lock:
            _check(self._lib.fz_seq_add_shard(seq._h, dev_index, addr, n, buf_global_off, own_lo, own_hi))
        del keep

    def device_ms(self):
        """Scan-kernel hipEvent span of the search collected last, one value per device of the engine."""
        n = len(self.devices)
        out = (ctypes.c_double * n)()
        with self._lock:                           # resolves the event spans: mutates the context
            rc = self._lib.fz_device_ms(self._h, out, n)
        if rc < 0:
            _raise(rc)
        return list(out)

    # -- RCCL without PyTorch (fz_comm_*) -----------------------------------------------------
    COMM_ID_BYTES = 128

    def comm_unique_id(self):
        """ncclGetUniqueId -> 128 bytes that rank 0 hands to the other ranks (file, socket, ...)."""
        buf = ctypes.create_string_buffer(self.COMM_ID_BYTES)
        _check(self._lib.fz_comm_unique_id(buf, self.COMM_ID_BYTES))
        return buf.raw

    def comm_init_rank(self, unique_id, world, rank):
        """One process per GPU: this (single-device) engine becomes rank `rank` of `world`.  From now on its
        lev_ngrams / lev_ngrams_begin / _end are COLLECTIVE and deliver the merged stream of all ranks."""
        if len(unique_id) != self.COMM_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % self.COMM_ID_BYTES)
        with self._lock:
            _check(self._lib.fz_comm_init_rank(self._h, ctypes.c_char_p(bytes(unique_id)), world, rank))

    def comm_init_all(self):
        """One process, several GPUs: every device of the engine becomes one rank (ncclCommInitAll)."""
        with self._lock:
            _check(self._lib.fz_comm_init_all(self._h))

    def comm_info(self):
        """-> (world, rank of device 0, collective searches on?); world == 0: no communicator."""
        w, r, c = ctypes.c_int(0), ctypes.c_int(-1), ctypes.c_int(0)
        _check(self._lib.fz_comm_info(self._h, ctypes.byref(w), ctypes.byref(r), ctypes.byref(c)))
        return w.value, r.value, bool(c.value)

    def comm_set_collective(self, on):
        with self._lock:
            _check(self._lib.fz_comm_set_collective(self._h, 1 if on else 0))

    def comm_allgather(self, data):
        """Equal-sized bytes from every rank -> list of bytes in rank order (ncclAllGather through device buffers)."""
        data = bytes(data)
        world = self.comm_info()[0]
        out = ctypes.create_string_buffer(max(1, world * len(data)))
        with self._lock:
            _check(self._lib.fz_comm_allgather(self._h, ctypes.c_char_p(data), len(data), out))
        raw = out.raw
        return [raw[i * len(data):(i + 1) * len(data)] for i in range(world)]

    def comm_max(self, value):
        v = ctypes.c_double(value)
        with self._lock:
            _check(self._lib.fz_comm_max_f64(self._h, ctypes.byref(v)))
        return v.value

    def comm_gather_ms(self):
        """Host milliseconds of the exchange step (all-gather + D2H + parse) of the collective search collected last."""
        v = ctypes.c_double(0.0)
        _check(self._lib.fz_comm_gather_ms(self._h, ctypes.byref(v)))
        return v.value

    @staticmethod
    def comm_backend():
        """'rccl', 'stand-in' (tests/mock_rccl.cpp through FZ_RCCL_LIB) or None when no collective library loads."""
        return {0: None, 1: "rccl", 2: "stand-in"}[load_library().fz_comm_backend()]

    def comm_barrier(self):
        with self._lock:
            _check(self._lib.fz_comm_barrier(self._h))

    def comm_destroy(self):
        with self._lock:
            if self._h is not None:
                self._lib.fz_comm_destroy(self._h)

    # -- searches (raw streams, tuples (start, end, dist, block)) ------------------------------
    def search_exact(self, seq, pattern, lo=0, hi=None, as_array=False):
        """Ascending start indices of every occurrence: a list of ints, or (as_array=True) a numpy int64 array."""
        import numpy as np
        paddr, m, keep = _buffer_address(pattern)
        ptr = ctypes.POINTER(ctypes.c_uint64)()
        cnt = ctypes.c_uint64(0)
        with self._lock:
            _check(self._lib.fz_search_exact(self._h, seq._h, paddr, m, max(0, lo),
                                             UINT64_MAX if hi is None else max(0, hi),
                                             ctypes.byref(ptr), ctypes.byref(cnt)))
        n = cnt.value
        arr = np.empty(n, dtype=np.int64)
        if n:
            ctypes.memmove(arr.ctypes.data, ptr, 8 * n)
        self._lib.fz_free(ptr)
        return arr if as_array else arr.tolist()

    def _match_call(self, fn, seq, pattern, *ints, **kw):
        if type(pattern) is bytes:                 # ctypes passes a bytes object as a pointer to its buffer
            paddr, m, keep = pattern, len(pattern), None
        else:
            paddr, m, keep = _buffer_address(pattern)
        ptr = ctypes.POINTER(FzMatch)()
        cnt = ctypes.c_uint64(0)
        with self._lock:
            _check(fn(self._h, seq._h, paddr, m, *ints, ctypes.byref(ptr), ctypes.byref(cnt)))
        if kw.get("as_array"):
            return _take_matches_array(self._lib, ptr, cnt.value)
        return _take_matches(self._lib, ptr, cnt.value)

    def rows_call(self, fn, seq, pattern, *ints):
        """One C-ABI search -> OwnedRows: the library's result buffer itself (address + row count), for callers that turn
        the rows into Match objects in C (common.matches_from_rows) — no numpy array, no second copy."""
        if type(pattern) is bytes:
            paddr, m, keep = pattern, len(pattern), None
        else:
            paddr, m, keep = _buffer_address(pattern)
        ptr = ctypes.POINTER(FzMatch)()
        cnt = ctypes.c_uint64(0)
        with self._lock:
            _check(fn(self._h, seq._h, paddr, m, *ints, ctypes.byref(ptr), ctypes.byref(cnt)))
        return OwnedRows(self._lib, ptr, cnt.value)

    def lev_ngrams_consolidated(self, seq, pattern, k, as_array=False):
        """consolidate_overlapping_matches(find_near_matches_levenshtein_ngrams(...)) in one call (fz_lev_ngrams_consolidated)."""
        return self._match_call(self._lib.fz_lev_ngrams_consolidated, seq, pattern, k, as_array=as_array)

    def subs_ngrams_best(self, seq, pattern, k, as_array=False):
        """Best match of every overlap group of the substitutions-only n-gram stream, group-list order (fz_subs_ngrams_best)."""
        return self._match_call(self._lib.fz_subs_ngrams_best, seq, pattern, k, as_array=as_array)

    def lev_ngrams(self, seq, pattern, k, as_array=False):
        """Raw stream of find_near_matches_levenshtein_ngrams: list of (start, end, dist, block)
        tuples, or a numpy structured array with those fields (as_array=True, no per-record
        Python objects)."""
        return self._match_call(self._lib.fz_lev_ngrams, seq, pattern, k, as_array=as_array)

    def lev_ngrams_begin(self, seq, pattern, k):
        """Launch lev_ngrams and return; lev_ngrams_end() delivers the result of the OLDEST search in flight.
        Up to two searches may be in flight per engine (the scan of search i + 1 runs while the host orders
        and consumes the records of search i); the host and other streams (a collective, a copy) can work
        meanwhile."""
        if type(pattern) is bytes:
            paddr, m, keep = pattern, len(pattern), None
        else:
            paddr, m, keep = _buffer_address(pattern)
        with self._lock:
            _check(self._lib.fz_lev_ngrams_begin(self._h, seq._h, paddr, m, k))

    def subs_ngrams_begin(self, seq, pattern, k):
        """subs_ngrams launched, not collected: search_end() delivers it (same two-deep pipeline as lev_ngrams_begin)."""
        paddr, m, keep = (pattern, len(pattern), None) if type(pattern) is bytes else _buffer_address(pattern)
        with self._lock:
            _check(self._lib.fz_subs_ngrams_begin(self._h, seq._h, paddr, m, k))

    def generic_ngrams_begin(self, seq, pattern, max_subs, max_ins, max_dels, max_l, consolidated=False):
        """generic_ngrams (or generic_ngrams_consolidated) launched, not collected.  Two generic searches in flight run
        on two lanes: the younger one's scan next to the older one's automaton kernel."""
        paddr, m, keep = (pattern, len(pattern), None) if type(pattern) is bytes else _buffer_address(pattern)
        with self._lock:
            _check(self._lib.fz_generic_ngrams_begin(self._h, seq._h, paddr, m, max_subs, max_ins, max_dels, max_l,
                                                     1 if consolidated else 0))

    def search_end(self, as_array=False):
        """The result of the OLDEST search in flight (lev / subs / generic begin), as its synchronous call returns it."""
        return self.lev_ngrams_end(as_array=as_array)

    def lev_ngrams_end(self, as_array=False):
        ptr = ctypes.POINTER(FzMatch)()
        cnt = ctypes.c_uint64(0)
        with self._lock:
            _check(self._lib.fz_lev_ngrams_end(self._h, ctypes.byref(ptr), ctypes.byref(cnt)))
        if as_array:
            return _take_matches_array(self._lib, ptr, cnt.value)
        return _take_matches(self._lib, ptr, cnt.value)

    def subs_ngrams(self, seq, pattern, k, as_array=False):
        return self._match_call(self._lib.fz_subs_ngrams, seq, pattern, k, as_array=as_array)

    def generic_ngrams(self, seq, pattern, max_subs, max_ins, max_dels, max_l, as_array=False):
        return self._match_call(self._lib.fz_generic_ngrams, seq, pattern, max_subs, max_ins, max_dels, max_l,
                                as_array=as_array)

    def generic_ngrams_consolidated(self, seq, pattern, max_subs, max_ins, max_dels, max_l, as_array=False):
        """consolidate_overlapping_matches(find_near_matches_generic_ngrams(...)) with the first stage of the consolidation
        on the device (fz_generic_ngrams_consolidated): rows (start, end, dist, block) sorted by (start, end, dist)."""
        return self._match_call(self._lib.fz_generic_ngrams_consolidated, seq, pattern, max_subs, max_ins, max_dels, max_l,
                                as_array=as_array)

    def lev_lp(self, seq, pattern, k, as_array=False):
        return self._match_call(self._lib.fz_lev_lp, seq, pattern, k, as_array=as_array)

    def subs_lp(self, seq, pattern, k, as_array=False):
        return self._match_call(self._lib.fz_subs_lp, seq, pattern, k, as_array=as_array)

    def generic_lp(self, seq, pattern, max_subs, max_ins, max_dels, max_l, as_array=False):
        return self._match_call(self._lib.fz_generic_lp, seq, pattern, max_subs, max_ins, max_dels, max_l,
                                as_array=as_array)

    def _any_call(self, fn, seq, pattern, *ints):
        paddr, m, keep = _buffer_address(pattern)
        found = ctypes.c_int(0)
        with self._lock:
            _check(fn(self._h, seq._h, paddr, m, *ints, ctypes.byref(found)))
        del keep
        return bool(found.value)

    def subs_ngrams_any(self, seq, pattern, k):
        """has_near_match_substitutions_ngrams: is the raw stream of subs_ngrams non-empty? (flag only, early exit)"""
        return self._any_call(self._lib.fz_subs_ngrams_any, seq, pattern, k)

    def subs_lp_any(self, seq, pattern, k):
        return self._any_call(self._lib.fz_subs_lp_any, seq, pattern, k)

    def generic_ngrams_any(self, seq, pattern, max_subs, max_ins, max_dels, max_l):
        return self._any_call(self._lib.fz_generic_ngrams_any, seq, pattern, max_subs, max_ins, max_dels, max_l)

    def set_timing(self, on):
        """hipEvent timing of the kernels (stats()["filter_ms"] ...): on by default; off saves a few us per call."""
        with self._lock:
            _check(self._lib.fz_set_timing(self._h, 1 if on else 0))

    def set_streams(self, n):
        """1 (default) or 2 streams for the two-deep pipeline of fused searches (fz_set_streams)."""
        with self._lock:
            _check(self._lib.fz_set_streams(self._h, n))

    def stats(self):
        st = FzStats()
        with self._lock:                           # fz_stats resolves the event spans: it mutates the context
            _check(self._lib.fz_stats(self._h, ctypes.byref(st)))
        return {f: getattr(st, f) for f, _ in FzStats._fields_}

    def mem_info(self):
        """(free, total) device memory in bytes: the minimum over the engine's devices."""
        f, t = ctypes.c_uint64(0), ctypes.c_uint64(0)
        with self._lock:
            _check(self._lib.fz_mem_info(self._h, ctypes.byref(f), ctypes.byref(t)))
        return f.value, t.value

    def kernel_ms(self):
        """(filter_ms, verify_ms, device_ms) of the last call: the cheap subset of stats()."""
        with self._lock:
            st = self._st
            _check(self._lib.fz_stats(self._h, self._st_ref))
            return st.filter_ms, st.verify_ms, st.device_ms


_default_engine = None
_default_lock = threading.Lock()


def default_engine():
    """Process-wide engine.  FUZZYSEARCH_HIP_DEVICES="0,1,..." selects the devices."""
    global _default_engine
    with _default_lock:
        if _default_engine is None:
            env = os.environ.get("FUZZYSEARCH_HIP_DEVICES", "").strip()
            devices = [int(x) for x in env.split(",") if x.strip()] if env else None
            _default_engine = Engine(devices)
        return _default_engine
