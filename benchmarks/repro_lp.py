import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native
import oracle
from tests import workloads
seq = workloads.dna(4 << 20, 78); pattern = workloads.dna(20, 1); workloads.plant_variants(seq, pattern, 256, 5)
p, t = pattern.tobytes(), seq.tobytes()
small = t[:1 << 16]; ps = p[:5]
want = oracle.lev_lp_raw(ps, small, 2)
eng = _native.Engine([0])
hs = eng.upload(small)
got = eng.lev_lp(hs, ps, 2)
print("plain: ok", got == want, len(got), len(want))
eng.comm_init_rank(eng.comm_unique_id(), 1, 0)
got = eng.lev_lp(hs, ps, 2)
print("collective: ok", got == want, len(got), len(want))
print("subs_lp", eng.subs_lp(hs, ps, 2) == oracle.subs_lp_raw(ps, small, 2), "generic_lp", eng.generic_lp(hs, ps, 1, 1, 1, 2) == oracle.generic_lp_raw(ps, small, 1, 1, 1, 2))
bad = [i for i in range(min(len(got), len(want))) if got[i] != want[i]][:3]
print("first diffs", bad, [(got[i], want[i]) for i in bad])
import collections
cg, cw = collections.Counter(got), collections.Counter(want)
print("missing", list((cw - cg).items())[:8], "extra", list((cg - cw).items())[:8])
