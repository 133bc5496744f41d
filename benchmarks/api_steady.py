"""configs[1] only, steady state (as bench.py times its `find_near_matches_ms`): the synchronous C-ABI search, the same with
the consolidation (fz_lev_ngrams_consolidated), and find_near_matches on a resident sequence — mean of `reps` calls after
0.2 s of warm-up.  The difference between the last two is Match objects + the Python layer; between the first two the
host's consolidation of the raw rows."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuzzysearch_amd as fa
from fuzzysearch_amd import _native
from tests import workloads

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = mib << 20
seq = workloads.dna(n, 20250925)
pat = workloads.dna(20, 1)
workloads.plant_variants(seq, pat, mib or 64, 7)
p = pat.tobytes()
res = fa.resident(seq.tobytes())
eng = _native.default_engine()


def steady(fn):
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        r = fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps * 1e3, r


raw_ms, raw = steady(lambda: eng.lev_ngrams(res.handle, p, 2, as_array=True))
cons_ms, cons = steady(lambda: eng.lev_ngrams_consolidated(res.handle, p, 2, as_array=True))
api_ms, out = steady(lambda: fa.find_near_matches(p, res, max_l_dist=2))
print(json.dumps({"case": "configs[1] DNA levenshtein k=2, steady state", "MiB": mib, "raw_matches": int(len(raw)), "result_matches": len(out),
                  "c_abi_ms": round(raw_ms, 4), "c_abi_consolidated_ms": round(cons_ms, 4), "find_near_matches_ms": round(api_ms, 4),
                  "api_over_c_abi": round(api_ms / raw_ms, 3), "match_type": fa.Match.__module__ + "." + type(out[0]).__name__ + (" (C)" if fa.Match is not getattr(fa.common, "_AttrsMatch", fa.Match) else " (attrs)")}))
