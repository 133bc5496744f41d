"""A/B timing of the BASELINE workloads on whatever libfzhip build FUZZYSEARCH_HIP_LIB names.
    FUZZYSEARCH_HIP_LIB=benchmarks/r1/libfzhip_r1.so python benchmarks/ab_scan.py [MiB] [reps]
Prints one JSON line per workload: C-ABI ms per call, scan / verify kernel ms (hipEvent), counts."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native  # noqa: E402
from tests import workloads  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = mib << 20
eng = _native.Engine([0])
tag = os.path.basename(_native.LIB_PATH)


def run(name, seq, call):
    h = eng.upload(seq)
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        call(h)
    fm, vm = [], []
    t0 = time.perf_counter()
    for _ in range(reps):
        r = call(h)
        f, v, _d = eng.kernel_ms()
        fm.append(f)
        vm.append(v)
    dt = (time.perf_counter() - t0) / reps
    st = eng.stats()
    print(json.dumps({"lib": tag, "workload": name, "MiB": mib, "ms_per_call": round(dt * 1e3, 4),
                      "GB_per_s": round(n / dt / 1e9, 1), "scan_ms": round(float(np.mean(fm)), 4),
                      "scan_ms_min": round(float(np.min(fm)), 4), "verify_ms": round(float(np.mean(vm)), 4),
                      "hits": st["ngram_hits"], "raw": len(r)}), flush=True)
    h.release()


seq, pat, _ = workloads.cfg2(n, 1024 * mib // 1024 or 64)
p = pat.tobytes()
run("cfg1 DNA m=20 k=2", seq, lambda h: eng.lev_ngrams(h, p, 2, as_array=True))
run("cfg0 DNA m=20 exact", seq, lambda h: eng.search_exact(h, p))
if "--all" in sys.argv:
    seq, pat, _ = workloads.cfg3(n, 1024 * mib // 1024 or 64)
    p2 = pat.tobytes()
    run("cfg2 ASCII m=32 subs<=3", seq, lambda h: eng.subs_ngrams(h, p2, 3, as_array=True))
    seq, pat, _ = workloads.cfg4(n, 1024 * mib // 1024 or 64)
    p3 = pat.tobytes()
    run("cfg3a UTF-8 m=64 k=5", seq, lambda h: eng.lev_ngrams(h, p3, 5, as_array=True))
    run("cfg3b UTF-8 m=64 (5,2,2,5)", seq, lambda h: eng.generic_ngrams(h, p3, 5, 2, 2, 5, as_array=True))
    run("cfg3b consolidated on the device", seq, lambda h: eng.generic_ngrams_consolidated(h, p3, 5, 2, 2, 5, as_array=True))
