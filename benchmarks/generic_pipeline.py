"""configs[3b] with one and two generic searches in flight (run on the GPU box):
    python benchmarks/generic_pipeline.py          (FZ_GEN_HI_STREAM=1: the automaton on a high-priority stream of its lane)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native
from tests import workloads
eng = _native.Engine([0])
seq, pat, _ = workloads.cfg4(1 << 30, 1024)
p = pat.tobytes()
h = eng.upload(seq)
want = eng.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True)
out = {"hi_stream": bool(os.environ.get("FZ_GEN_HI_STREAM"))}
for cons in (False, True):
    t_end = time.perf_counter() + 0.3
    fn = (lambda: eng.generic_ngrams_consolidated(h, p, 5, 2, 2, 5, as_array=True)) if cons else (lambda: eng.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True))
    while time.perf_counter() < t_end:
        fn()
    t0 = time.perf_counter()
    for _ in range(60):
        fn()
    out["sync_ms_%s" % ("consolidated" if cons else "raw")] = round((time.perf_counter() - t0) / 60 * 1e3, 4)
    eng.generic_ngrams_begin(h, p, 5, 2, 2, 5, consolidated=cons)
    for _ in range(10):
        eng.generic_ngrams_begin(h, p, 5, 2, 2, 5, consolidated=cons)
        eng.search_end(as_array=True)
    t0 = time.perf_counter()
    for _ in range(100):
        eng.generic_ngrams_begin(h, p, 5, 2, 2, 5, consolidated=cons)
        r = eng.search_end(as_array=True)
    out["two_in_flight_ms_%s" % ("consolidated" if cons else "raw")] = round((time.perf_counter() - t0) / 100 * 1e3, 4)
    last = eng.search_end(as_array=True)
    if not cons:
        assert np.array_equal(r, want) and np.array_equal(last, want)
print(json.dumps(out), flush=True)
