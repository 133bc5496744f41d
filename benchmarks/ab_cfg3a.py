"""Levenshtein budgets 5 .. 8 on 1 GiB (configs[3a]: UTF-8 text, m = 64; and a candidate-dense case: DNA, m = 40, k = 5):
C-ABI time, kernel spans, two searches in flight, a digest of the rows — for the verification form the environment selects
(FZ_NO_WF_FUSE=1: the stand-alone lane-per-cell kernel instead of the fused one)."""
import hashlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native
from tests import workloads
eng = _native.Engine([0])


def run(tag, h, p, k, reps=200):
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        r = eng.lev_ngrams(h, p, k, as_array=True)
    f, v = [], []
    t0 = time.perf_counter()
    for _ in range(reps):
        r = eng.lev_ngrams(h, p, k, as_array=True)
        a, b, _d = eng.kernel_ms(); f.append(a); v.append(b)
    dt = (time.perf_counter() - t0) / reps
    hits = eng.stats()["ngram_hits"]
    eng.lev_ngrams_begin(h, p, k)
    for _ in range(10):
        eng.lev_ngrams_begin(h, p, k); eng.lev_ngrams_end(as_array=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.lev_ngrams_begin(h, p, k); eng.lev_ngrams_end(as_array=True)
    pipe = (time.perf_counter() - t0) / reps
    eng.lev_ngrams_end(as_array=True)
    print(json.dumps({"case": tag, "fused_wf": "FZ_NO_WF_FUSE" not in os.environ, "wf32": os.environ.get("FZ_WF32", "adaptive"), "k": k, "ms_per_call": round(dt * 1e3, 4),
                      "two_in_flight_ms": round(pipe * 1e3, 4), "scan_ms": round(float(np.mean(f)), 4),
                      "verify_ms": round(float(np.mean(v)), 4), "raw": len(r), "hits": hits,
                      "sha": hashlib.sha1(r.tobytes()).hexdigest()[:12]}), flush=True)


seq, pat, _ = workloads.cfg4(1 << 30, 1024)
h = eng.upload(seq)
for k in (5, 6, 7, 8, 12):
    run("utf8 m=64", h, pat.tobytes(), k)
run("utf8 m=64, pattern of bytes the text does not hold", h, bytes(range(1, 65)), 5)
h.release()
seq = workloads.dna(1 << 30, 77)
pat = workloads.dna(40, 5)
workloads.plant_variants(seq, pat, 1024, 9)
h = eng.upload(seq)
run("dna m=40", h, pat.tobytes(), 5, reps=30)
run("dna m=48", h, workloads.dna(48, 6).tobytes(), 5, reps=50)
run("dna m=54", h, workloads.dna(54, 7).tobytes(), 8, reps=20)
run("dna m=100", h, workloads.dna(100, 8).tobytes(), 10, reps=50)
