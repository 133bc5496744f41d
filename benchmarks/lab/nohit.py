"""Timing of the headline scan instance without candidates: DNA text, a pattern of letters DNA does not have (m = 20, k = 2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fuzzysearch_amd import _native
from tests import workloads
n = 1 << 30
eng = _native.Engine([0])
seq = workloads.dna(n, 5)
pat = workloads.text65(20, 6)
h = eng.upload(seq); p = pat.tobytes()
t_end = time.perf_counter() + 0.4
while time.perf_counter() < t_end: eng.lev_ngrams(h, p, 2, as_array=True)
fms = []
for _ in range(300):
    r = eng.lev_ngrams(h, p, 2, as_array=True); fms.append(eng.kernel_ms()[0])
st = eng.stats()
print("%s no-hit headline instance: scan %.4f ms (min %.4f) hits %d recs %d" % (os.path.basename(_native.LIB_PATH), np.mean(fms), np.min(fms), st["ngram_hits"], len(r)))
