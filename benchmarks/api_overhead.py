"""Public API against the C-ABI at large result counts: BASELINE configs[3b] — 1 GiB of UTF-8
text, m = 64, limits (5, 2, 2, 5), 2.1e5 raw matches — and configs[1].  find_near_matches() on a resident()
sequence = C-ABI search + consolidation on the array in C++ + Match objects for the survivors only."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuzzysearch_amd as fa
from fuzzysearch_amd import _native
from tests import workloads

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r


seq, pat, _ = workloads.cfg4(n, 1024 * mib // 1024 or 64)
p = pat.tobytes()
res = fa.resident(seq)
eng = _native.default_engine()
ms_abi, raw = best(lambda: eng.generic_ngrams(res.handle if hasattr(res, "handle") else res._handle, p, 5, 2, 2, 5, as_array=True))
ms_api, out = best(lambda: fa.find_near_matches(p, res, max_substitutions=5, max_insertions=2, max_deletions=2, max_l_dist=5))
ms_cons, rows = best(lambda: eng.generic_ngrams_consolidated(res.handle, p, 5, 2, 2, 5, as_array=True))
print(json.dumps({"case": "configs[3b] generic (5,2,2,5)", "MiB": mib, "raw_matches": int(len(raw)), "result_matches": len(out),
                  "c_abi_ms": round(ms_abi, 3), "c_abi_consolidated_ms": round(ms_cons, 3), "consolidated_rows": int(len(rows)),
                  "find_near_matches_ms": round(ms_api, 3), "ratio": round(ms_api / ms_abi, 2)}), flush=True)
ms_abi, raw = best(lambda: eng.lev_ngrams(res.handle if hasattr(res, "handle") else res._handle, p, 5, as_array=True))
ms_api, out = best(lambda: fa.find_near_matches(p, res, max_l_dist=5))
print(json.dumps({"case": "configs[3a] levenshtein k=5", "MiB": mib, "raw_matches": int(len(raw)), "result_matches": len(out),
                  "c_abi_ms": round(ms_abi, 3), "find_near_matches_ms": round(ms_api, 3), "ratio": round(ms_api / ms_abi, 2)}), flush=True)
res.release()
# configs[2]: substitutions-only (group-list-order reduction, fz_group_best) and configs[1]: DNA, k = 2
seq, pat, _ = workloads.cfg3(n, 1024 * mib // 1024 or 64)
p = pat.tobytes()
res = fa.resident(seq)
ms_abi, raw = best(lambda: eng.subs_ngrams(res.handle, p, 3, as_array=True))
ms_api, out = best(lambda: fa.find_near_matches(p, res, max_substitutions=3, max_insertions=0, max_deletions=0))
print(json.dumps({"case": "configs[2] substitutions <= 3", "MiB": mib, "raw_matches": int(len(raw)), "result_matches": len(out),
                  "c_abi_ms": round(ms_abi, 3), "find_near_matches_ms": round(ms_api, 3), "ratio": round(ms_api / ms_abi, 2)}), flush=True)
res.release()
seq = workloads.dna(n, 20250925)
pat = workloads.dna(20, 1)
workloads.plant_variants(seq, pat, 1024 * mib // 1024 or 64, 7)
p = pat.tobytes()
res = fa.resident(seq)
ms_abi, raw = best(lambda: eng.lev_ngrams(res.handle, p, 2, as_array=True))
ms_api, out = best(lambda: fa.find_near_matches(p, res, max_l_dist=2))
print(json.dumps({"case": "configs[1] DNA levenshtein k=2", "MiB": mib, "raw_matches": int(len(raw)), "result_matches": len(out),
                  "c_abi_ms": round(ms_abi, 3), "find_near_matches_ms": round(ms_api, 3), "ratio": round(ms_api / ms_abi, 2)}), flush=True)
