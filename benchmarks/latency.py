"""What one synchronous search costs beyond its kernel (run on the GPU box):
    python benchmarks/latency.py [MiB ...]
For every size: ms per fz_lev_ngrams call at the C-ABI (through ctypes, numpy result) with the hipEvent timing on
(kernel ms known) and off (the leaner product path), the headline pattern.  FZ_TRACE=1 adds the host phases."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native  # noqa: E402
from tests import workloads  # noqa: E402

sizes = [int(x) for x in sys.argv[1:] if x.isdigit()] or [1, 1024]
eng = _native.Engine([0])
for mib in sizes:
    seq, pat, _ = workloads.cfg2(mib << 20, max(16, 1024 * mib // 1024))
    p = pat.tobytes()
    h = eng.upload(seq)
    out = {"MiB": mib}
    for timing in (True, False, True, False):
        eng.set_timing(timing)
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            r = eng.lev_ngrams(h, p, 2, as_array=True)
        reps = 300
        km = []
        t0 = time.perf_counter()
        for _ in range(reps):
            r = eng.lev_ngrams(h, p, 2, as_array=True)
            if timing:
                km.append(eng.kernel_ms()[0])
        dt = (time.perf_counter() - t0) / reps * 1e3
        key = "timing_on" if timing else "timing_off"
        out.setdefault(key, []).append(round(dt, 4))
        if timing:
            out.setdefault("kernel_ms", []).append(round(float(np.mean(km)), 4))
    out["raw"] = len(r)
    print(json.dumps(out), flush=True)
    h.release()
