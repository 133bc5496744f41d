"""CPU-only: what the public API adds on the host to its C-ABI search — find_near_matches(p, resident, max_l_dist=2) with the
device call replaced by a canned result buffer of configs[1]'s size (1 026 consolidated rows), so that the Python layer
(parameters, strategy choice, prepare, ctypes call frame) and the Match objects can be timed without a GPU.
    python benchmarks/api_host_cost.py [rows]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fuzzysearch_amd as fz
from fuzzysearch_amd import _native, common, engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1026
rows = np.zeros(n, dtype=_native._match_dtype())
rows['start'] = np.sort(np.random.default_rng(1).integers(0, (1 << 24) - 64, n))
rows['end'] = rows['start'] + 20
rows['dist'] = np.arange(n) % 3
seq_bytes = bytes(1 << 24)


class _Lib(object):
    fz_lev_ngrams_consolidated = "fz_lev_ngrams_consolidated"

    @staticmethod
    def fz_free(ptr):
        pass


class _Engine(object):
    _lib = _Lib

    def rows_call(self, fn, seq, pattern, *ints):
        # the frame of _native.Engine.rows_call without the library call itself
        ptr = ctypes.cast(rows.ctypes.data, ctypes.POINTER(_native.FzMatch))
        cnt = ctypes.c_uint64(n)
        return _native.OwnedRows(self._lib, ptr, cnt.value)


res = object.__new__(engine.DeviceSequence)
res.original, res.engine, res.byteslike, res.handle = seq_bytes, _Engine(), True, object()
pat = b"ACGTACGTACGTACGTACGT"


def timed(fn, reps=3000):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


whole = timed(lambda: fz.find_near_matches(pat, res, max_l_dist=2))
fill = timed(lambda: common._fzmatch.make_matches(rows, seq_bytes, 0)) if common._fzmatch else float('nan')
n_save, rows_small = n, rows[:0]
layer = None
print("find_near_matches on a canned buffer of %d rows: %.1f us per call; Match objects (build + free of the previous list) %.1f us; "
      "the rest (parameters, strategy, prepare, call frame) %.1f us" % (n, whole, fill, whole - fill))
