"""Levenshtein linear-programming route (fz_lp_kernel<FZ_LP_LEV_SEQ, *>) against the oracle, for A/B runs of kernel
builds:   FUZZYSEARCH_HIP_LIB=benchmarks/lab/libfzhip_<name>.so python benchmarks/repro_lp.py
Prints one line per case family and the first differences."""
import collections
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from fuzzysearch_amd import _native  # noqa: E402
from tests import workloads  # noqa: E402

eng = _native.Engine([0])
seq = workloads.dna(4 << 20, 78)
pattern = workloads.dna(20, 1)
workloads.plant_variants(seq, pattern, 256, 5)
p, t = pattern.tobytes(), seq.tobytes()
bad_total = 0


def one(ps, text, k, label):
    global bad_total
    want = oracle.lev_lp_raw(ps, text, k)
    hs = eng.upload(text)
    got = eng.lev_lp(hs, ps, k)
    hs.release()
    ok = got == want
    if not ok:
        bad_total += 1
        cg, cw = collections.Counter(got), collections.Counter(want)
        print("  %s: MISMATCH got %d want %d missing %r extra %r" % (label, len(got), len(want), list((cw - cg).items())[:4], list((cg - cw).items())[:4]), flush=True)
    return ok, len(want)


ok, n = one(p[:5], t[:1 << 16], 2, "dna 64 KiB m=5 k=2")
print("dna 64 KiB m=5 k=2:", ok, n, flush=True)
ok, n = one(p[:7], t[:1 << 20], 3, "dna 1 MiB m=7 k=3")
print("dna 1 MiB m=7 k=3:", ok, n, flush=True)
rnd = random.Random(5)
n_ok = n_rows = 0
N = int(os.environ.get("REPRO_CASES", "300"))
for i in range(N):
    alpha = bytes(rnd.sample(range(1, 256), rnd.choice([2, 3, 4, 20])))
    k = rnd.choice([1, 2, 3])
    m = rnd.randint(2, 3 * (k + 1) - 1)
    nn = rnd.randint(0, 3000)
    text = bytes(rnd.choice(alpha) for _ in range(nn))
    ps = bytes(rnd.choice(alpha) for _ in range(m))
    ok, n = one(ps, text, k, "random %d (n=%d m=%d k=%d sigma=%d)" % (i, nn, m, k, len(alpha)))
    n_ok += ok
    n_rows += n
print("random: %d / %d ok, %d rows; families with differences: %d" % (n_ok, N, n_rows, bad_total), flush=True)
