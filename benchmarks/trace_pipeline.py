"""Under `rocprofv3 --kernel-trace`: the two-deep pipeline of the headline search (1 GiB and 4 GiB DNA) — the kernel trace's
start / end stamps give the gaps between consecutive scan kernels (benchmarks/trace_pipeline.sh summarises them)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fuzzysearch_amd import _native
from tests import workloads
eng = _native.Engine([0])
pat = workloads.dna(20, 1); p = pat.tobytes()
for mib in (1024, 4096):
    seq = np.empty(mib << 20, dtype=np.uint8)
    for i in range(mib >> 10):
        seq[i << 30:(i + 1) << 30] = workloads.dna(1 << 30, 20250925 + i)
    workloads.plant_variants(seq, pat, mib, 7)
    h = eng.upload(seq)
    for _ in range(30):
        eng.lev_ngrams(h, p, 2, as_array=True)
    eng.lev_ngrams_begin(h, p, 2)
    t0 = time.perf_counter()
    for _ in range(200):
        eng.lev_ngrams_begin(h, p, 2)
        eng.lev_ngrams_end(as_array=True)
    dt = (time.perf_counter() - t0) / 200
    eng.lev_ngrams_end(as_array=True)
    print("mib %d: %.4f ms per step (two in flight)" % (mib, dt * 1e3), flush=True)
    h.release()
    del seq
