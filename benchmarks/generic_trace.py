"""Where one synchronous configs[3b] call spends its time on the host (FZ_TRACE marks of libfzhip, stderr):
    python benchmarks/generic_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native
from tests import workloads
eng = _native.Engine([0])
seq, pat, _ = workloads.cfg4(1 << 30, 1024)
p = pat.tobytes()
h = eng.upload(seq)
for name, fn in (("consolidated", lambda: eng.generic_ngrams_consolidated(h, p, 5, 2, 2, 5, as_array=True)),
                 ("raw", lambda: eng.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True)),
                 ("lev k=5", lambda: eng.lev_ngrams(h, p, 5, as_array=True))):
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        fn()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    print("%s: %.4f ms per call, stats %r" % (name, (time.perf_counter() - t0) / 50 * 1e3, eng.stats()), file=sys.stderr, flush=True)
    os.environ["FZ_TRACE"] = "1"
    _native.load_library().fz_debug_reload_switches()       # (the library reads its switches once)
    for _ in range(3):
        print("--", name, file=sys.stderr, flush=True)
        fn()
    del os.environ["FZ_TRACE"]
    _native.load_library().fz_debug_reload_switches()
