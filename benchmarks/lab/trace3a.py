"""FZ_TRACE=1 split of one synchronous call: configs[1] against configs[3a] (where the microseconds outside the kernel go)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fuzzysearch_amd import _native
from tests import workloads
n = 1 << 30
eng = _native.Engine([0])
for name, mk, k in (("cfg1", workloads.cfg2, 2), ("cfg3a", workloads.cfg4, 5)):
    seq, pat, _ = mk(n, 1024)
    p = pat.tobytes()
    h = eng.upload(seq)
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end: eng.lev_ngrams(h, p, k, as_array=True)
    ts, ks, ds = [], [], []
    for _ in range(300):
        t0 = time.perf_counter(); r = eng.lev_ngrams(h, p, k, as_array=True); ts.append(time.perf_counter() - t0)
        f, v, d = eng.kernel_ms(); ks.append(f); ds.append(d)
    print("%s: call %.1f us, kernel %.1f us, device span (first launch -> results on host) %.1f us, records %d" % (name, np.mean(ts) * 1e6, np.mean(ks) * 1e3, np.mean(ds) * 1e3, len(r)), flush=True)
    sys.stderr.write("---- %s ----\n" % name); sys.stderr.flush()
    os.environ["FZ_TRACE"] = "1"; eng._lib.fz_debug_reload_switches()
    for _ in range(3): eng.lev_ngrams(h, p, k, as_array=True)
    os.environ["FZ_TRACE"] = "0"; eng._lib.fz_debug_reload_switches()
    h.release()
