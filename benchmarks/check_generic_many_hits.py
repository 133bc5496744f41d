"""Generic search with more n-gram hits than the device-side ordering takes (FZ_GEN_ORDER_MAX = 16384): the host
orders the records (their `win` field then holds the hit slot, not a segment); stream against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from fuzzysearch_amd import _native
rng = np.random.default_rng(5)
t = np.frombuffer(b'ACGT', np.uint8)[rng.integers(0, 4, 3 << 20, dtype=np.uint8)].tobytes()
p = t[1000:1012]
eng = _native.Engine([0])
h = eng.upload(t)
for lim in ((1, 1, 1, 2), (2, 0, 1, 2)):
    got = eng.generic_ngrams(h, p, *lim)
    st = eng.stats()
    t0 = time.time()
    exp = oracle.generic_ngrams_raw(p, t, *lim)
    print({"limits": lim, "ngram_hits": int(st["ngram_hits"]), "raw": len(got), "equal": got == exp, "oracle_s": round(time.time() - t0, 2)}, flush=True)
    assert got == exp
