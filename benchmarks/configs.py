"""BASELINE.json configs[1..3] (+4b) on one MI355X: GB/s at the C-ABI with the sequence resident,
each checked against the oracle on the full input (raw streams bit-exact).  Run on the GPU box:
    python benchmarks/configs.py [MiB]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from fuzzysearch_amd import _native  # noqa: E402
from tests import workloads  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20
eng = _native.Engine([0])


utf8_text = workloads.utf8_text


def timeit(fn, reps=100, warm=60):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps, r


rows = []
cases = [
    ("cfg0 DNA m=20 exact (search_exact, max_l_dist=0)", lambda: workloads.cfg2(n, 1024 * mib // 1024 or 64)[:2],
     lambda h, p: eng.search_exact(h, p), lambda p, t: oracle.search_exact(p, t)),
    ("cfg1 DNA m=20 k=2 (levenshtein_ngram)", lambda: workloads.cfg2(n, 1024 * mib // 1024 or 64)[:2],
     lambda h, p: eng.lev_ngrams(h, p, 2, as_array=True), lambda p, t: oracle.lev_ngrams_raw(p, t, 2)),
    ("cfg2 ASCII m=32 subs<=3 (substitutions_only)", None,
     lambda h, p: eng.subs_ngrams(h, p, 3, as_array=True), lambda p, t: oracle.subs_ngrams_raw(p, t, 3)),
    ("cfg3a UTF-8 m=64 k=5 (levenshtein_ngram, wide band)", None,
     lambda h, p: eng.lev_ngrams(h, p, 5, as_array=True), lambda p, t: oracle.lev_ngrams_raw(p, t, 5)),
    ("cfg3b UTF-8 m=64 (5,2,2,5) (generic_search)", None,
     lambda h, p: eng.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True), lambda p, t: oracle.generic_ngrams_raw(p, t, 5, 2, 2, 5)),
]
for name, gen, run, orc in cases:
    if gen is not None:
        seq, pat = gen()
    elif "ASCII" in name:
        seq, pat, _ = workloads.cfg3(n, 1024 * mib // 1024 or 64)
    else:
        seq, pat, _ = workloads.cfg4(n, 1024 * mib // 1024 or 64)
    h = eng.upload(seq)
    p = pat.tobytes()
    dt, res = timeit(lambda: run(h, p))
    st = eng.stats()
    t0 = time.perf_counter()
    exp = orc(p, seq.tobytes())
    t_cpu = time.perf_counter() - t0
    res = res.tolist() if hasattr(res, "tolist") else res
    got = [tuple(int(x) for x in r) if isinstance(r, (tuple, list)) else int(r) for r in res]
    ok = got == exp
    rows.append({"config": name, "MiB": mib, "ms_per_call": round(dt * 1e3, 4), "GB_per_s": round(n / dt / 1e9, 1),
                 "scan_kernel_ms": round(st["filter_ms"], 4), "kernel_GB_per_s": round(n / st["filter_ms"] / 1e6, 1),
                 "verify_ms": round(st["verify_ms"], 4), "ngram_hits": st["ngram_hits"], "raw_matches": len(got),
                 "bit_exact_vs_oracle": ok, "oracle_C_port_seconds": round(t_cpu, 2)})
    print(json.dumps(rows[-1]), flush=True)
    h.release()
    assert ok, name
