"""Substitutions-only n-gram search where candidates are dense (DNA, short patterns): ms per call at the C-ABI."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native
from tests import workloads
eng = _native.default_engine()
seq = workloads.dna(int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 1 << 30, 20250925)
h = eng.upload(seq)
only = os.environ.get("ONLY")
for m, k in ([tuple(int(x) for x in only.split(","))] if only else [(32, 3), (20, 2), (20, 3), (20, 4), (12, 2), (12, 3)]):
    p = workloads.dna(m, int(os.environ.get("PSEED", "1"))).tobytes()
    r = eng.subs_ngrams(h, p, k, as_array=True)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 0.3 or n < 3:
        r = eng.subs_ngrams(h, p, k, as_array=True); n += 1
    ms = (time.perf_counter() - t0) / n * 1e3
    st = eng.stats()
    print("subs m=%d k=%d: %.4f ms  kernel %.4f  candidates %d rows %d" % (m, k, ms, eng.kernel_ms()[0], st["ngram_hits"], len(r)), flush=True)
