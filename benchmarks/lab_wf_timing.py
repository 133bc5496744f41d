"""Lab (library built with -DFZ_LAB_LPTIME -DFZ_LAB_SCANTIME): phases of the scan kernel's workgroups, 100 MHz stamps of
wave 0 of every workgroup: 0 kernel entry, 2 tables ready, 1 final flush (after the pool barrier), 3 flush done, 4 before
the finish tickets, 7 end.  argv: k [pattern: text|absent]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fuzzysearch_amd import _native
from tests import workloads
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
absent = len(sys.argv) > 2 and sys.argv[2] == "absent"
eng = _native.Engine([0])
seq, pat, _ = workloads.cfg4(1 << 30, 1024)
p = bytes(range(1, 65)) if absent else pat.tobytes()
h = eng.upload(seq)
for _ in range(20):
    r = eng.lev_ngrams(h, p, k, as_array=True)
f, v, _d = eng.kernel_ms()
L = _native.load_library()
buf = np.zeros(16384 * 4, dtype=np.uint64)
L.fz_lab_lp_read.restype = ctypes.c_int
assert L.fz_lab_lp_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_uint64(len(buf))) == 0
t = buf.reshape(-1, 8).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = lambda x: [round(float(y) / 100.0, 2) for y in np.percentile(x, [0, 10, 50, 90, 100])]
out = {"k": k, "absent": absent, "fused_wf": "FZ_NO_WF_FUSE" not in os.environ, "rows": len(r), "scan_ms": round(f, 4), "verify_ms": round(v, 4),
       "workgroups": int(len(t)),
       "entry_us": us(t[:, 0] - t0), "tables_us": us(t[:, 2] - t[:, 0]), "scan_us": us(t[:, 1] - t[:, 2]) if t[:, 1].max() > 0 else None,
       "flush_us": us(t[:, 3] - t[:, 1]) if t[:, 1].max() > 0 else None,
       "to_finish_us": us(t[:, 4] - np.where(t[:, 3] > 0, t[:, 3], t[:, 2])), "finish_us": us(t[:, 7] - t[:, 4]),
       "end_us": us(t[:, 7] - t0)}
print(json.dumps(out), flush=True)
