"""-m gpu: stress of the three linear-programming routes (fz_lev_lp / fz_subs_lp / fz_generic_lp; the reference takes them
when len(subsequence) // (max_l_dist + 1) < 3: levenshtein.py:52-148, substitutions_only.py:82-136,
generic_search.py:57-177) — >= 10 000 random cases and 1 MiB inputs with >= 10^5 matches, candidate lists in LDS and
(FZ_CAND_LDS_MAX=1, a subprocess: the knob is process-wide) in HBM, all against the oracle.

Why: round 4 shipped fz_lp_kernel<FZ_LP_LEV_SEQ> with a known-unexplained discrepancy in a sibling build ("slot form").
Round 5 root-caused it — a hipcc miscompile of a lane-divergent `break` loop (fz_device.h: fz_levlp_step_slots) — and
the kernel now runs the slot form with a wave-uniform loop exit; this test is what would have caught the broken form
(its first wrong case is within the first 300 of these) and guards the next compiler."""
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases(seed, n_cases, max_n):
    rnd = random.Random(seed)
    for i in range(n_cases):
        alpha = bytes(rnd.sample(range(1, 256), rnd.choice([2, 3, 4, 4, 20])))
        k = rnd.choice([1, 1, 2, 2, 3, 4])
        m = rnd.randint(1 if k > 1 else 2, 3 * (k + 1) - 1)               # m // (k + 1) < 3: the reference's LP regime
        n = rnd.choice([0, 1, m - 1, m, m + 1]) if rnd.random() < 0.05 else rnd.randint(0, max_n)
        text = bytearray(rnd.choice(alpha) for _ in range(max(0, n)))
        p = bytes(rnd.choice(alpha) for _ in range(m))
        if n > 4 * m and rnd.random() < 0.7:                               # copies at both ends and around the 256-start tile seams
            for at in (0, n - m, 256 - m // 2, 512 - 1, rnd.randint(0, n - m)):
                if 0 <= at <= n - m:
                    text[at:at + m] = p
        yield p, bytes(text), k


def run(n_cases, big, seed=20250926):
    """-> (cases, rows) checked."""
    import oracle
    from fuzzysearch_amd import _native
    eng = _native.default_engine()
    n_rows = n_done = 0
    for (p, t, k) in _cases(seed, n_cases, 1500):
        h = eng.upload(t)
        m = len(p)
        got = eng.lev_lp(h, p, k)
        assert got == oracle.lev_lp_raw(p, t, k), ("lev_lp", p, t, k)
        n_rows += len(got)
        ks = min(k, m)
        got = eng.subs_lp(h, p, ks)
        assert got == oracle.subs_lp_raw(p, t, ks), ("subs_lp", p, t, ks)
        n_rows += len(got)
        lim = (min(k, 2), min(k, 1), min(k, 2), k)
        got = eng.generic_lp(h, p, *lim)
        assert got == oracle.generic_lp_raw(p, t, *lim), ("generic_lp", p, t, lim)
        n_rows += len(got)
        h.release()
        n_done += 1
    if big:
        biggest = 0
        for seed_, sigma, m, k in ((1, 4, 7, 3), (2, 4, 5, 2), (3, 3, 8, 3), (4, 20, 4, 2)):
            rnd = random.Random(seed_)
            alpha = bytes(rnd.sample(range(65, 91), sigma))
            t = bytes(alpha[b % sigma] for b in rnd.randbytes(1 << 20))
            p = bytes(rnd.choice(alpha) for _ in range(m))
            h = eng.upload(t)
            want = oracle.lev_lp_raw(p, t, k)
            got = eng.lev_lp(h, p, k)
            assert got == want, ("lev_lp 1 MiB", p, k, len(got), len(want))
            n_rows += len(got)
            biggest = max(biggest, len(want))
            want = oracle.generic_lp_raw(p, t, k, 1, 1, k)
            got = eng.generic_lp(h, p, k, 1, 1, k)
            assert got == want, ("generic_lp 1 MiB", p, k, len(got), len(want))
            n_rows += len(got)
            want = oracle.subs_lp_raw(p, t, min(k, 2))
            assert eng.subs_lp(h, p, min(k, 2)) == want
            n_rows += len(want)
            h.release()
            n_done += 3
        assert biggest >= 100000, biggest                                  # at least one 1 MiB input with >= 10^5 matches
    return n_done, n_rows


def test_lp_routes_ten_thousand_random_cases_lists_in_lds():
    n_done, n_rows = run(10000, big=True)
    assert n_done >= 10000 and n_rows > 2000000


def test_lp_routes_lists_in_hbm():
    """The same routes with the candidate lists forced into HBM (fz_lp_kernel<*, true>): 1 500 random cases + the 1 MiB inputs."""
    env = dict(os.environ)
    env["FZ_CAND_LDS_MAX"] = "1"
    code = "from tests import test_gpu_lp_stress as t; print('DONE %d %d' % t.run(1500, big=True, seed=7))"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    done = [ln for ln in out.stdout.splitlines() if ln.startswith("DONE ")]
    assert done and int(done[-1].split()[1]) >= 1500 and int(done[-1].split()[2]) > 500000, out.stdout[-500:]


if __name__ == "__main__":
    print("checked %d cases, %d rows" % run(int(sys.argv[1]) if len(sys.argv) > 1 else 2000, big="big" in sys.argv))
