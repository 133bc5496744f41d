import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fuzzysearch_amd import _native
from tests import workloads
eng = _native.Engine([0])
seq, pat, _ = workloads.cfg4(1 << 30, 1024); p = pat.tobytes(); h = eng.upload(seq)
for _ in range(10): r = eng.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True)
st = eng.stats(); L = _native.load_library()
buf = np.zeros(16384 * 4, dtype=np.uint64); L.fz_lab_lp_read.restype = ctypes.c_int
assert L.fz_lab_lp_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_uint64(len(buf))) == 0
t = buf.reshape(-1, 4)[:int(st["ngram_hits"])]
ran = t[:, 2] > 0
step = (t[ran, 0] >> np.uint64(32)).astype(np.int64); scan = (t[ran, 0] & np.uint64(0xffffffff)).astype(np.int64)
store = (t[ran, 1] >> np.uint64(32)).astype(np.int64); head = (t[ran, 1] & np.uint64(0xffffffff)).astype(np.int64)
trips = ((t[ran, 3] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64)
big = trips >= 60
for name, v in (("step", step), ("scan", scan), ("store", store), ("head/loop", head)):
    print("%-10s per trip (hits with >= 60 trips): median %.0f cycles" % (name, np.median(v[big] / trips[big])))
print("sum per trip %.0f; hits %d, big %d, automaton_ms %.4f" % (np.median((step + scan + store + head)[big] / trips[big]), ran.sum(), big.sum(), st["verify_ms"]))
