"""configs[3a] only (UTF-8 text, m = 64, k = 5): scan + verify kernel times for the library FUZZYSEARCH_HIP_LIB names."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native
from tests import workloads
eng = _native.Engine([0])
seq, pat, _ = workloads.cfg4(1 << 30, 1024)
p = pat.tobytes()
h = eng.upload(seq)
for k in (5, 7, 8):
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        r = eng.lev_ngrams(h, p, k, as_array=True)
    f, v = [], []
    t0 = time.perf_counter()
    for _ in range(200):
        r = eng.lev_ngrams(h, p, k, as_array=True)
        a, b, _d = eng.kernel_ms(); f.append(a); v.append(b)
    dt = (time.perf_counter() - t0) / 200
    print(json.dumps({"lib": os.path.basename(_native.LIB_PATH), "k": k, "ms_per_call": round(dt * 1e3, 4), "scan_ms": round(float(np.mean(f)), 4),
                      "verify_ms": round(float(np.mean(v)), 4), "verify_min": round(float(np.min(v)), 4), "raw": len(r), "hits": eng.stats()["ngram_hits"]}), flush=True)
