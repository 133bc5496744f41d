"""configs[3b] only (UTF-8 text, m = 64, limits (5,2,2,5)): C-ABI ms per call, scan and automaton kernel times."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native
from tests import workloads
eng = _native.Engine([0])
seq, pat, _ = workloads.cfg4(1 << 30, 1024)
p = pat.tobytes()
h = eng.upload(seq)
cons = bool(os.environ.get("FZ_AB_CONSOLIDATED"))          # fz_generic_ngrams_consolidated instead of the raw stream
call = (lambda: eng.generic_ngrams_consolidated(h, p, 5, 2, 2, 5, as_array=True)) if cons else (lambda: eng.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True))
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    r = call()
f, v, dv = [], [], []
t0 = time.perf_counter()
for _ in range(60):
    r = call()
    a, b, _d = eng.kernel_ms(); f.append(a); v.append(b); dv.append(_d)
dt = (time.perf_counter() - t0) / 60
print(json.dumps({"lib": os.path.basename(_native.LIB_PATH), "consolidated": cons, "grid_per_cu": os.environ.get("FZ_LP_GRID_PER_CU", "16"), "ms_per_call": round(dt * 1e3, 4),
                  "scan_ms": round(float(np.mean(f)), 4), "automaton_ms": round(float(np.mean(v)), 4), "device_ms": round(float(np.mean(dv)), 4),
                  "order": "host" if os.environ.get("FZ_GEN_HOST_ORDER") else "device", "sha": __import__("hashlib").sha1(r.tobytes()).hexdigest()[:12], "raw": len(r)}), flush=True)
