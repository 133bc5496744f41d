"""Where the microseconds of one synchronous fz_lev_ngrams call go (run on the GPU box): the raw ctypes call, the numpy
hand-over, the record ordering alone (fz_debug_order_records on records like the search's), np.array_equal."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native  # noqa: E402
from tests import workloads  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = _native.Engine([0])
lib = eng._lib
seq, pat, _ = workloads.cfg2(mib << 20, 1024)
p = pat.tobytes()
h = eng.upload(seq)


def per_call(fn, reps=300, warm=0.3):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end:
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


ptr = ctypes.POINTER(_native.FzMatch)()
cnt = ctypes.c_uint64(0)


def raw_call():
    lib.fz_lev_ngrams(eng._h, h._h, p, len(p), 2, ctypes.byref(ptr), ctypes.byref(cnt))
    lib.fz_free(ptr)


res = eng.lev_ngrams(h, p, 2, as_array=True)
for timing in (True, False):
    eng.set_timing(timing)
    print("timing %s: raw ctypes call + fz_free %.1f us; Engine.lev_ngrams(as_array) %.1f us; kernel %.1f us"
          % (timing, per_call(raw_call), per_call(lambda: eng.lev_ngrams(h, p, 2, as_array=True)), eng.kernel_ms()[0] * 1e3))
eng.set_timing(True)
other = res.copy()
print("np.array_equal of two %d-row results: %.1f us" % (len(res), per_call(lambda: np.array_equal(res, other), 2000, 0.05)))
rec_dt = np.dtype([("key", "<u8"), ("l", "<u4"), ("r", "<u4"), ("dist", "<u4"), ("aux", "<u4")])
rng = np.random.default_rng(1)
recs = np.zeros(len(res), rec_dt)
recs["key"] = (rng.integers(0, 3, len(res)).astype(np.uint64) << np.uint64(48)) | rng.integers(0, mib << 20, len(res)).astype(np.uint64)


def order():
    lib.fz_debug_order_records(recs.ctypes.data, len(recs), 6, ctypes.byref(ptr), ctypes.byref(cnt))
    lib.fz_free(ptr)


print("record ordering alone (%d records, cached): %.1f us" % (len(recs), per_call(order, 2000, 0.05)))
