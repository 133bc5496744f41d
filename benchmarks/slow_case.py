"""One pathological stress case, call by call with timings (which search of stress_parity.py takes minutes?)."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from fuzzysearch_amd import _native
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
alpha = b"abcd"
n, m, k = 500, 16, 12
t = bytes(rnd.choice(alpha) for _ in range(n)); p = bytes(rnd.choice(alpha) for _ in range(m))
eng = _native.Engine([0]); h = eng.upload(t)
def timed(name, fn, want=None):
    t0 = time.time()
    try:
        r = fn()
        ok = "" if want is None else (" equal" if r == want() else " DIFFERENT")
        print("%-28s %8.2f s  %d rows%s" % (name, time.time() - t0, len(r) if hasattr(r, "__len__") else -1, ok), flush=True)
    except NotImplementedError as e:
        print("%-28s %8.2f s  UnsupportedSearch: %s" % (name, time.time() - t0, str(e)[:80]), flush=True)
timed("lev_ngrams", lambda: eng.lev_ngrams(h, p, k), lambda: oracle.lev_ngrams_raw(p, t, k))
timed("subs_ngrams", lambda: eng.subs_ngrams(h, p, k), lambda: oracle.subs_ngrams_raw(p, t, k))
for lim in ((6, 7, 3, 12), (12, 12, 12, 12), (2, 1, 1, 12)):
    timed("generic_ngrams %r" % (lim,), lambda: eng.generic_ngrams(h, p, *lim), lambda: oracle.generic_ngrams_raw(p, t, *lim))
    timed("generic_cons %r" % (lim,), lambda: eng.generic_ngrams_consolidated(h, p, *lim))
timed("lev_lp", lambda: eng.lev_lp(h, p, k), lambda: oracle.lev_lp_raw(p, t, k))
timed("subs_lp", lambda: eng.subs_lp(h, p, k), lambda: oracle.subs_lp_raw(p, t, k))
timed("generic_lp (6,7,3,12)", lambda: eng.generic_lp(h, p, 6, 7, 3, 12), lambda: oracle.generic_lp_raw(p, t, 6, 7, 3, 12))
