import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fuzzysearch_amd import _native
from tests import workloads
mib, m, k, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
n = mib << 20
eng = _native.Engine([0])
alpha = workloads.DNA if kind == "dna" else workloads.TEXT65
seq = workloads.dna(n, 5) if kind == "dna" else workloads.text65(n, 5)
pat = workloads.dna(m, 6) if kind == "dna" else workloads.text65(m, 6)
workloads.plant_variants(seq, pat, 1024, 7, alpha)
h = eng.upload(seq); p = pat.tobytes()
for _ in range(60): eng.lev_ngrams(h, p, k, as_array=True)
t0 = time.perf_counter(); N = 100
fms = []
for _ in range(N):
    r = eng.lev_ngrams(h, p, k, as_array=True); fms.append(eng.stats()["filter_ms"])
dt = (time.perf_counter() - t0) / N
st = eng.stats()
print("m=%d k=%d %s %d MiB: %.4f ms/call; scan %.4f ms (%.0f GB/s) verify %.4f; hits %d recs %d" % (m, k, kind, mib, dt*1e3, np.mean(fms), n/np.mean(fms)/1e6, st["verify_ms"], st["ngram_hits"], len(r)))
