"""The verification regimes of the Levenshtein n-gram search on one GiB of DNA (the cliff map of VERDICT r05 item 1):
|p| = 20 with k = 1, 2, 3, 4; |p| = 54, k = 8; |p| = 100, k = 20 — ms per call at the C-ABI, scan / verify kernel ms,
n-gram hits (= verified candidates) and raw matches.  `bench.py` prints the same block; this script is for A/B runs.

    python benchmarks/regimes.py [--mib 1024] [--reps 10] [--check]      (--check: 64 MiB against the oracle first)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

REGIMES = [(20, 1), (20, 2), (20, 3), (20, 4), (54, 8), (100, 20)]


def run(engine, seq, reps, regimes=REGIMES, warm_s=0.2):
    from tests import workloads
    out = []
    h = engine.upload(seq)
    for m, k in regimes:
        p = workloads.dna(m, 7 if m != 20 else 1).tobytes()
        t0 = time.perf_counter()
        first = engine.lev_ngrams(h, p, k, as_array=True)
        first_ms = (time.perf_counter() - t0) * 1e3
        t_end = time.perf_counter() + warm_s                           # clocks settle (as bench.py's time_call)
        while time.perf_counter() < t_end:
            engine.lev_ngrams(h, p, k, as_array=True)
        f_ms, v_ms = [], []
        t0 = time.perf_counter()
        for _ in range(reps):
            res = engine.lev_ngrams(h, p, k, as_array=True)
            f_, v_, _d = engine.kernel_ms()
            f_ms.append(f_)
            v_ms.append(v_)
        ms = (time.perf_counter() - t0) / reps * 1e3
        st = engine.stats()
        assert np.array_equal(first, res)
        out.append({"m": m, "k": k, "ms": round(ms, 4), "GB_per_s": round(len(seq) / ms / 1e6, 1), "first_ms": round(first_ms, 3),
                    "scan_kernel_ms": round(float(np.mean(f_ms)), 4), "verify_kernel_ms": round(float(np.mean(v_ms)), 4),
                    "candidates": int(st["ngram_hits"]), "raw_matches": int(len(res))})
    h.release()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", default="", help="m,k[;m,k...]: only these regimes")
    args = ap.parse_args()
    from fuzzysearch_amd import _native
    from tests import workloads
    engine = _native.default_engine()
    if args.check:
        import oracle
        seq = workloads.dna(8 << 20, 99)
        for m, k in REGIMES:
            p = workloads.dna(m, 7 if m != 20 else 1)
            workloads.plant_edits(seq, p, 64, 5 + m, workloads.DNA, lambda i: i % (k + 1))
        h = engine.upload(seq)
        for m, k in REGIMES:
            p = workloads.dna(m, 7 if m != 20 else 1).tobytes()
            got = engine.lev_ngrams(h, p, k)
            exp = oracle.lev_ngrams_raw(p, seq.tobytes(), k)
            print("check m=%d k=%d: %d rows, equal=%s" % (m, k, len(exp), got == exp), flush=True)
            assert got == exp
        h.release()
    seq = workloads.dna(args.mib << 20, 20250925)
    workloads.plant_variants(seq, workloads.dna(20, 1), 1024, 7)
    regimes = [tuple(int(x) for x in r.split(",")) for r in args.only.split(";") if r] or REGIMES
    for row in run(engine, seq, args.reps, regimes):
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
