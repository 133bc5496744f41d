"""configs[3b] (1 GiB UTF-8 text, m = 64, limits (5, 2, 2, 5)): the generic search's C-ABI time and kernel spans for the
automaton forms selected by the environment (FZ_GEN_LEGACY=1, FZ_GH_WAVES=4, FZ_GEN_NO_DEDUP=1, FZ_GH_GRID_PER_CU)."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fuzzysearch_amd import _native
from tests import workloads
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = _native.Engine([0])
seq, pat, _ = workloads.cfg4(mib << 20, max(64, mib))
p = pat.tobytes()
h = eng.upload(seq)
def bench(fn, reps=100):
    t_end = time.perf_counter() + 0.3
    r = fn()
    while time.perf_counter() < t_end: r = fn()
    f, v = [], []
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn(); a, b, _d = eng.kernel_ms(); f.append(a); v.append(b)
    return (time.perf_counter() - t0) / reps * 1e3, float(np.mean(f)), float(np.mean(v)), r
out = {"env": {k: os.environ[k] for k in os.environ if k.startswith("FZ_G")}, "MiB": mib}
ms, f, v, r = bench(lambda: eng.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True))
out["raw"] = {"ms": round(ms, 4), "scan_ms": round(f, 4), "automaton_ms": round(v, 4), "rows": int(len(r)), "sha": hashlib.sha1(r.tobytes()).hexdigest()[:12]}
ms, f, v, r = bench(lambda: eng.generic_ngrams_consolidated(h, p, 5, 2, 2, 5, as_array=True))
out["consolidated"] = {"ms": round(ms, 4), "scan_ms": round(f, 4), "automaton_ms": round(v, 4), "rows": int(len(r)), "sha": hashlib.sha1(r.tobytes()).hexdigest()[:12]}
eng.generic_ngrams_begin(h, p, 5, 2, 2, 5)
for _ in range(10):
    eng.generic_ngrams_begin(h, p, 5, 2, 2, 5); eng.search_end(as_array=True)
t0 = time.perf_counter()
for _ in range(100):
    eng.generic_ngrams_begin(h, p, 5, 2, 2, 5); eng.search_end(as_array=True)
out["two_in_flight_ms"] = round((time.perf_counter() - t0) / 100 * 1e3, 4)
eng.search_end(as_array=True)
print(json.dumps(out), flush=True)
