import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native
import oracle
p, t, lim = b'cd.', b'djddcjcccjddjjjjdddddjjcdjddjddjc.jj.cc.cdddc.dccjjjddjcjcj.c..d.jdcc.c.c.cjjcj', (0, 1, 2, 2)
eng = _native.Engine([0])
h = eng.upload(t)
raw = eng.generic_ngrams(h, p, *lim)
want = oracle.generic_ngrams_raw(p, t, *lim)
print(dict((k, os.environ[k]) for k in os.environ if k.startswith("FZ_G")), "raw ok", raw == want, len(raw))
cons = eng.generic_ngrams_consolidated(h, p, *lim)
wc = oracle.consolidate(want)
print(" cons ok", [r[:3] for r in cons] == wc, len(cons), len(wc))
if [r[:3] for r in cons] != wc:
    print(" missing", [r for r in wc if r not in [c[:3] for c in cons]], "extra", [c for c in cons if c[:3] not in wc])
    print(" raw rows near 36:", [r for r in want if 30 <= r[0] <= 40])
