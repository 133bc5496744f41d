"""Where does a fz_lev_ngrams() step spend its host time?  (run on the GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fuzzysearch_amd import _native
from tests import workloads

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seq, pattern, _ = workloads.cfg2(mib << 20, 1024 * mib // 1024 or 256)
eng = _native.Engine([0])
h = eng.upload(seq)
p = pattern.tobytes()
for name, fn in [("tuples", lambda: eng.lev_ngrams(h, p, 2)), ("array", lambda: eng.lev_ngrams(h, p, 2, as_array=True))]:
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    N = 50
    for _ in range(N):
        r = fn()
    dt = (time.perf_counter() - t0) / N
    st = eng.stats()
    print("%-7s %d MiB: %.4f ms/call wall; device_ms %.4f filter_ms %.4f; %d records" % (name, mib, dt * 1e3, st["device_ms"], st["filter_ms"], len(r)))
