"""Lab (library built with -DFZ_LAB_LPTIME -DFZ_LAB_SCANTIME): what the workgroups of ONE scan launch do and when — 100 MHz
stamps of thread 0 of every workgroup (0 entry, 1 tables ready, 2 tiles done, 3 end-of-life flush done, 4 end) plus the
hardware id it ran on.  argv: workload (dna20k2 | exact | utf8k5) [MiB]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fuzzysearch_amd import _native
from tests import workloads
what = sys.argv[1] if len(sys.argv) > 1 else "dna20k2"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eng = _native.Engine([0])
if what == "utf8k5":
    seq, pat, _ = workloads.cfg4(mib << 20, mib)
    k = 5
else:
    seq, pat, _ = workloads.cfg2(mib << 20, mib)
    k = 0 if what == "exact" else 2
p = pat.tobytes()
h = eng.upload(seq)
call = (lambda: eng.search_exact(h, p, as_array=True)) if what == "exact" else (lambda: eng.lev_ngrams(h, p, k, as_array=True))
for _ in range(20):
    r = call()
f, v, _d = eng.kernel_ms()
L = _native.load_library()
buf = np.zeros(16384 * 4, dtype=np.uint64)
L.fz_lab_lp_read.restype = ctypes.c_int
assert L.fz_lab_lp_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_uint64(len(buf))) == 0
raw = buf.reshape(-1, 8)
raw = raw[raw[:, 0] > 0]
t = raw[:, :7].astype(np.int64)
hw = raw[:, 7]
t0 = t[:, 0].min()
US = 100.0
pc = lambda x: [round(float(y) / US, 2) for y in np.percentile(x, [0, 10, 50, 90, 100])]
entry, end = (t[:, 0] - t0) / US, (t[:, 4] - t0) / US
life = end - entry
xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
cu = (hwid >> 8) & 15
se = (hwid >> 13) & 7
sh = (hwid >> 12) & 1
place = xcc * 1024 + se * 64 + sh * 32 + cu
out = {"workload": what, "MiB": mib, "rows": int(len(r)), "scan_ms": round(f, 4), "workgroups": int(len(t)),
       "tables_us": pc(t[:, 1] - t[:, 0]), "tiles_us": pc(t[:, 2] - t[:, 1]), "flush_us": pc(t[:, 3] - t[:, 2]), "finish_us": pc(t[:, 4] - t[:, 3]),
       "life_us": pc(t[:, 4] - t[:, 0]), "kernel_span_us": round(float(end.max()), 2),
       "first_round_workgroups": int((entry < 2.0).sum()), "places": int(len(set(place.tolist()))), "xccs": sorted(set(xcc.tolist()))}
# residency over time, and how the launch ends
edges = np.arange(0, end.max() + 10, 10.0)
out["resident_every_10us"] = [int(((entry <= x) & (end > x)).sum()) for x in edges]
last_start = float(entry.max())
out["last_workgroup_starts_us"] = round(last_start, 2)
out["tail_after_last_start_us"] = round(float(end.max()) - last_start, 2)
out["life_by_start_quartile_us"] = [round(float(np.median(life[(entry >= a) & (entry <= b)])), 2) for a, b in
                                     zip(np.percentile(entry, [0, 25, 50, 75]), np.percentile(entry, [25, 50, 75, 100]))]
# per place (CU): workgroups served, when the last one ended
ends = {}
cnt = {}
for pl, e in zip(place.tolist(), end.tolist()):
    ends[pl] = max(ends.get(pl, 0.0), e)
    cnt[pl] = cnt.get(pl, 0) + 1
ev = np.array(list(ends.values()))
cv = np.array(list(cnt.values()))
out["per_cu_last_end_us"] = pc(ev * US)
out["per_cu_workgroups"] = [int(x) for x in np.percentile(cv, [0, 10, 50, 90, 100])]
xe = {}
for x, e in zip(xcc.tolist(), end.tolist()):
    xe[x] = max(xe.get(x, 0.0), e)
out["per_xcc_last_end_us"] = {str(k2): round(v2, 1) for k2, v2 in sorted(xe.items())}
print(json.dumps(out), flush=True)
