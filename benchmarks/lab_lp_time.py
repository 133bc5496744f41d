"""Lab (library built with -DFZ_LAB_LPTIME): where the generic search's automaton kernel spends its time, hit by hit —
shader-clock stamps at start / window staged / end of every n-gram hit of configs[3b]."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fuzzysearch_amd import _native
from tests import workloads
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = _native.Engine([0])
seq, pat, _ = workloads.cfg4(mib << 20, max(64, mib))
p = pat.tobytes()
h = eng.upload(seq)
for _ in range(20):
    r = eng.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True)
st = eng.stats()
L = _native.load_library()
buf = np.zeros(16384 * 4, dtype=np.uint64)
L.fz_lab_lp_read.restype = ctypes.c_int
assert L.fz_lab_lp_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_uint64(len(buf))) == 0
t = buf.reshape(-1, 4)
nh = int(st["ngram_hits"])
t = t[:nh]
ran = t[:, 2] > 0
t0 = t[ran, 0].min()
beg, stg, end = (t[ran, 0] - t0).astype(np.int64), (t[ran, 1] - t0).astype(np.int64), (t[ran, 2] - t0).astype(np.int64)
dur = end - beg
slices = (t[ran, 3] >> np.uint64(32)).astype(np.int64)
out = {"env": {k: os.environ[k] for k in os.environ if k.startswith("FZ_G")}, "hits": nh, "ran": int(ran.sum()), "verify_ms": st["verify_ms"],
       "start_cycles_pct": [int(x) for x in np.percentile(beg, [0, 50, 90, 99, 100])],
       "stage_cycles_pct": [int(x) for x in np.percentile(stg - beg, [0, 50, 90, 99, 100])],
       "dur_cycles_pct": [int(x) for x in np.percentile(dur, [0, 10, 50, 90, 99, 100])],
       "end_cycles_pct": [int(x) for x in np.percentile(end, [0, 50, 90, 99, 100])],
       "slowest": [[int(x) for x in (dur[i], stg[i] - beg[i], slices[i])] for i in np.argsort(dur)[-5:]],
       "slices_pct": [int(x) for x in np.percentile(slices, [0, 50, 90, 100])] if slices.max() else None,
       "cycles_per_slice_step_median": float(np.median(dur[slices > 0] / slices[slices > 0])) if slices.max() else None}
print(json.dumps(out), flush=True)
