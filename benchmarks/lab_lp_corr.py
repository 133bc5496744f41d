"""Lab (-DFZ_LAB_LPTIME): per-hit duration of fz_gen_hit_kernel against the hit's candidate steps, trips and matches."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fuzzysearch_amd import _native
from tests import workloads
eng = _native.Engine([0])
seq, pat, _ = workloads.cfg4(1 << 30, 1024)
p = pat.tobytes()
h = eng.upload(seq)
for _ in range(10):
    r = eng.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True)
st = eng.stats()
L = _native.load_library()
buf = np.zeros(16384 * 4, dtype=np.uint64)
L.fz_lab_lp_read.restype = ctypes.c_int
assert L.fz_lab_lp_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_uint64(len(buf))) == 0
t = buf.reshape(-1, 4)[:int(st["ngram_hits"])]
ran = t[:, 2] > 0
dur = (t[ran, 2] - t[ran, 0]).astype(np.int64)
stage = (t[ran, 1] - t[ran, 0]).astype(np.int64)
cands = (t[ran, 3] >> np.uint64(32)).astype(np.int64)
trips = ((t[ran, 3] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64)
mb = (t[ran, 3] & np.uint64(0xffff)).astype(np.int64)
order = np.argsort(dur)
print("n", len(dur), "corr dur~cands %.3f dur~trips %.3f dur~matches %.3f" % (np.corrcoef(dur, cands)[0, 1], np.corrcoef(dur, trips)[0, 1], np.corrcoef(dur, mb)[0, 1]))
for name, idx in (("fastest", order[:6]), ("median", order[len(order) // 2 - 3:len(order) // 2 + 3]), ("slowest", order[-6:])):
    print(name, [(int(dur[i]), int(stage[i]), int(cands[i]), int(trips[i]), int(mb[i])) for i in idx])
A = np.stack([cands, trips, mb, np.ones_like(cands)], axis=1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, dur.astype(np.float64), rcond=None)
print("least squares: dur ~ %.2f * cands + %.1f * trips + %.1f * matches + %.0f" % tuple(coef))
