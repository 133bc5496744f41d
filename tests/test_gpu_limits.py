"""-m gpu: the edges of what the engine accepts (the reference has no limits: levenshtein_ngram.py:159-198).
Long subsequences (pattern in HBM), budgets beyond the LDS ring, hundreds to a thousand n-gram blocks (dozens of
scan launches), fz_verify_big_kernel at every cells-per-lane width, the documented refusals through the C-ABI, the
process-wide switches in subprocesses, and a slice of the randomized stress run."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import oracle
from tests import gpu_cases, workloads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sub(args, env, timeout=600):
    e = dict(os.environ)
    e.update(env)
    res = subprocess.run([sys.executable, "-m", "tests.gpu_cases"] + [str(a) for a in args], cwd=ROOT, env=e,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = res.stdout.decode()
    assert res.returncode == 0 and "OK " in out, out[-3000:]
    return [int(x) for x in out.strip().splitlines()[-1].split()[1:]]


def test_big_verify_kernel_forced_on_random_cases():
    n_cases, n_rec = _sub(["big", 500, 7], {"FZ_FORCE_BIG_VERIFY": "1"})
    assert n_cases == 500 and n_rec > 500


def test_general_slot_form_and_multi_launch():
    n_cases, n_rec = _sub(["slots", 300, 9], {"FZ_NO_SLOT_AND": "1"})
    assert n_cases == 302 and n_rec > 1000
    n_cases, n_rec = _sub(["slots", 200, 10], {"FZ_MAX_BLOCKS": "2"})
    assert n_cases == 202


def test_generic_window_table_on_and_off():
    """Round 4: hits that share their window run the automaton once (smallest block = leader, the others take its rows).
    The same random cases with the table (default) and without (FZ_GEN_NO_DEDUP=1) against the oracle: identical
    streams either way."""
    n_cases, n_rec = _sub(["windows", 250, 21], {})
    assert n_cases == 250 and n_rec > 5000
    n_cases2, n_rec2 = _sub(["windows", 250, 21], {"FZ_GEN_NO_DEDUP": "1"})
    assert (n_cases2, n_rec2) == (n_cases, n_rec)


def test_generic_automaton_kernel_forms():
    """The per-hit automaton in every form the library has — round 5's default for patterns up to 64 characters and budgets up
    to 32: the bit-parallel step, ONE wave per hit (fz_gen_hit_kernel<1, true>: 64-bit equality words, flags as words,
    unconditional stores, starts that cannot reach the pattern's end not spawned); the same step with the window's starts
    dealt out to 2 or 4 waves whose sorted match buffers are merged by rank (FZ_GH_WAVES); round 4's step on 2 waves
    (FZ_GH_NO_BITS=1, also what longer patterns / larger budgets take); one wave per hit through fz_lp_kernel
    (FZ_GEN_LEGACY=1) — the same random cases (dense repeats among them: hits that outgrow a wave's match buffer make the
    search run again on fz_lp_kernel, with a back-off for the searches that follow; patterns beyond 64 characters among
    them) against the oracle: raw stream, consolidated rows incl. the block, flag, two in flight."""
    base = _sub(["windows", 200, 23], {})
    assert base[0] == 200 and base[1] > 4000
    small = _sub(["windows", 100, 25], {})
    assert small[0] == 100 and small[1] > 2000
    assert _sub(["windows", 100, 25], {"FZ_GH_NO_BITS": "1"}) == small
    assert _sub(["windows", 100, 25], {"FZ_GH_WAVES": "2"}) == small
    assert _sub(["windows", 100, 25], {"FZ_GH_WAVES": "4"}) == small
    assert _sub(["windows", 100, 25], {"FZ_GEN_LEGACY": "1"}) == small
    assert _sub(["windows", 60, 24], {"FZ_GH_WAVES": "4", "FZ_GEN_NO_DEDUP": "1"})[0] == 60


def test_scan_grid_regions_leave_the_streams_alone():
    """Round 4: the last resident round of a big scan launch takes shrinking tile shares (FzScanArgs.reg_*).  768 MiB of DNA
    — Levenshtein k = 2 (fused), exact (hit list), substitutions, k = 5 with 36 bytes (six-byte n-grams, lane-per-cell
    verification in the scan: a million candidates) — without regions, with the default taper and with a steep one: the same digests."""
    base = _sub(["taper", 768], {"FZ_TAPER_STEPS": "0"})
    assert base[0] == 768 and base[1] > 1500
    assert _sub(["taper", 768], {}) == base
    assert _sub(["taper", 768], {"FZ_TAPER_STEPS": "7", "FZ_TAPER_MIN": "0.05", "FZ_TAPER_WG_PER_CU": "10"}) == base


def test_lane_per_cell_verification_fused_and_stand_alone():
    """Levenshtein budgets 5 .. 15 of an in-memory search verify inside the scan kernel (fz_flush_wf: the queued candidates,
    four at a time on 16 lanes each up to budget 7, two at a time on 32 lanes beyond); FZ_NO_WF_FUSE=1 sends the same
    searches through the hit list and fz_verify_wf_kernel.  The same random cases — ragged ends, long patterns, alphabets small enough to fill the queues —
    against the oracle either way."""
    base = _sub(["wf", 300, 31], {})                  # budgets 8 .. 15: fused or not by the density the previous search saw
    assert base[0] == 312 and base[1] > 1000
    assert _sub(["wf", 300, 31], {"FZ_WF32": "1"}) == base        # ... always fused
    assert _sub(["wf", 300, 31], {"FZ_NO_WF_FUSE": "1"}) == base  # nothing fused


def test_copy_mode_one_and_two_searches_in_flight():
    """FZ_NO_DIRECT=1: nothing is written straight into the pinned staging buffer (round 3's randomized run found the
    younger of two searches in flight overwriting the older one's records in the shared device buffer under this switch)."""
    n_cases, n_rec = _sub(["copy", 120, 11], {"FZ_NO_DIRECT": "1"})
    assert n_cases == 123 and n_rec > 100000
    n_cases, n_rec = _sub(["copy", 40, 12], {})               # the same cases in the default (direct) mode
    assert n_cases == 43 and n_rec > 100000


def _planted(rnd, n, p, alpha, edits_list, corrupt_blocks=None, L=None):
    """Random text over alpha with edited copies of p; corrupt_blocks = (n_blocks, L): instead of random edits, one
    substitution in each of n_blocks randomly chosen n-gram blocks (so that exactly the others hit)."""
    t = bytearray(rnd.choices(alpha, k=n))
    pos = 50
    for e in edits_list:
        if corrupt_blocks is None:
            v = gpu_cases.edited(rnd, p, e, alpha)
        else:
            v = bytearray(p)
            G = len(p) // L
            for g in rnd.sample(range(G), e):
                q = g * L + rnd.randrange(L)
                v[q] = rnd.choice([c for c in alpha if c != v[q]])
            v = bytes(v)
        if pos + len(v) + 50 > n:
            break
        t[pos:pos + len(v)] = v
        pos += len(v) + rnd.randint(30, 200)
    return bytes(t)


@pytest.mark.parametrize("m,k,sigma,n,edits", [
    (1000, 3, 4, 60000, [0, 1, 3, 3, 2]),            # L = 250: the exact re-check walks long n-grams
    (1020, 3, 4, 40000, [0, 2, 3]),
    (1024, 3, 4, 40000, [3, 1]),                     # the last size that travels in the kernel arguments
    (1025, 3, 4, 40000, [3, 0, 2]),                  # the first that goes through HBM
    (2000, 40, 4, 30000, [0, 17, 40]),               # G = 41 blocks, band 81: two cells per lane
    (1000, 200, 20, 6000, [150, 200]),               # L = 4, G = 250, band 401: eight cells per lane
    (1024, 255, 20, 6000, [255, 100]),               # L = 4, G = 256
    (255, 254, 20, 400, [100]),                      # L = 1, G = 255: every byte of the text hits some block
    (4096, 80, 4, 30000, [0, 80, 33]),               # L = 50, G = 81
])
def test_long_patterns_and_wide_budgets_levenshtein(engine, m, k, sigma, n, edits):
    rnd = random.Random(m * 1000 + k)
    alpha = bytes(rnd.sample(range(1, 256), sigma))
    p = bytes(rnd.choices(alpha, k=m))
    t = _planted(rnd, n, p, alpha, edits)
    h = engine.upload(t)
    got = engine.lev_ngrams(h, p, k)
    st = engine.stats()
    h.release()
    exp = oracle.lev_ngrams_raw(p, t, k)
    assert got == exp
    assert len(exp) >= 1 and st["filter_launches"] >= (m // (m // (k + 1)) + 15) // 16


def test_m_16384_k_300(engine):
    """The verdict's size: m = 16 384, k = 300 (L = 54, G = 303 blocks, 38+ scan launches, band of 601 cells = 16
    per lane).  One planted copy with a substitution in 290 of its blocks: 13 blocks hit, each is verified over the
    whole 16 KiB pattern (the oracle's full DP needs a few seconds for them)."""
    rnd = random.Random(16384)
    alpha = bytes(rnd.sample(range(1, 256), 4))
    m, k = 16384, 300
    L = m // (k + 1)
    p = bytes(rnd.choices(alpha, k=m))
    t = _planted(rnd, 60000, p, alpha, [290], corrupt_blocks=True, L=L)
    h = engine.upload(t)
    got = engine.lev_ngrams(h, p, k)
    h.release()
    exp = oracle.lev_ngrams_raw(p, t, k)
    assert got == exp
    assert len(exp) >= 10 and min(d for (_s, _e, d, _g) in exp) == 290


def test_1000_blocks(engine):
    """m = 3000, k = 999: L = 3, G = 1000 blocks (at least 63 launches of up to 16), band of 1999 cells = 32 per lane."""
    rnd = random.Random(3000)
    alpha = bytes(rnd.sample(range(1, 256), 90))
    m, k = 3000, 999
    p = bytes(rnd.choices(alpha, k=m))
    t = _planted(rnd, 9000, p, alpha, [900], corrupt_blocks=True, L=3)
    h = engine.upload(t)
    got = engine.lev_ngrams(h, p, k)
    st = engine.stats()
    h.release()
    exp = oracle.lev_ngrams_raw(p, t, k)
    assert got == exp
    assert len(exp) >= 90 and max(g for (_s, _e, _d, g) in exp) > 255 and st["filter_launches"] >= 63


def test_long_patterns_other_routes(engine):
    rnd = random.Random(77)
    alpha = bytes(rnd.sample(range(1, 256), 4))
    # substitutions-only: m = 5000, k = 40 (L = 121, G = 41): big Hamming verification
    p = bytes(rnd.choices(alpha, k=5000))
    t = bytearray(rnd.choices(alpha, k=40000))
    for off, nsub in ((100, 0), (9000, 40), (20000, 41), (30000, 17)):
        v = bytearray(p)
        for q in rnd.sample(range(5000), nsub):
            v[q] = rnd.choice([c for c in alpha if c != v[q]])
        t[off:off + 5000] = v
    t = bytes(t)
    h = engine.upload(t)
    assert engine.subs_ngrams(h, p, 40) == oracle.subs_ngrams_raw(p, t, 40)
    assert engine.subs_lp(h, p[:1500], 3) == oracle.subs_lp_raw(p[:1500], t, 3)
    # exact search of a 16 KiB needle
    needle = t[100:100 + 4000] + bytes(rnd.choices(alpha, k=12384))
    t2 = t[:35000] + needle + t[35000:] + needle[:-1]
    h2 = engine.upload(t2)
    assert engine.search_exact(h2, needle) == oracle.search_exact(needle, t2) == [35000]
    h2.release()
    # generic search: m = 2000, limits (3, 1, 1, 3): L = 500
    pg = t[9000:11000]
    got = engine.generic_ngrams(h, pg, 3, 1, 1, 3)
    assert got == oracle.generic_ngrams_raw(pg, t, 3, 1, 1, 3) and len(got) > 0
    h.release()


def test_refusals_through_the_c_abi(engine):
    from fuzzysearch_amd import _native
    t = workloads.dna(1 << 16, 1).tobytes()
    h = engine.upload(t)
    with pytest.raises(_native.UnsupportedSearch):
        engine.lev_ngrams(h, t[:65536], 3)                       # > 65 535 items
    with pytest.raises(_native.UnsupportedSearch):
        engine.lev_ngrams(h, t[:40000], 1024)                    # budget above 1023
    with pytest.raises(_native.UnsupportedSearch):
        engine.generic_ngrams(h, t[:1000], 256, 256, 256, 256)   # the automaton's counters are 8 bits wide
    # the automaton's records carry window-relative coordinates in 16 bits: m + 2k (generic n-gram route) and
    # m + 2k + 256 (linear-programming tiles) must fit, else the search is refused (round 3 silently wrapped)
    with pytest.raises(_native.UnsupportedSearch):
        engine.generic_ngrams(h, t[:65500], 20, 20, 20, 20)
    with pytest.raises(_native.UnsupportedSearch):
        engine.generic_ngrams_consolidated(h, t[:65500], 20, 20, 20, 20)
    with pytest.raises(_native.UnsupportedSearch):
        engine.generic_ngrams_any(h, t[:65500], 20, 20, 20, 20)
    with pytest.raises(_native.UnsupportedSearch):
        engine.lev_lp(h, t[:65300], 2)
    with pytest.raises(_native.UnsupportedSearch):
        engine.generic_lp(h, t[:65300], 2, 2, 2, 2)
    with pytest.raises(ValueError):
        engine.lev_ngrams(h, t[:5], 5)                           # n-gram length 0 (levenshtein_ngram.py:164-165)
    assert engine.lev_ngrams(h, t[:65535], 2)[0][:3] == (0, 65535, 0)
    h.release()


def test_generic_route_at_the_16_bit_window_limit(engine):
    """m + 2k = 65 408 <= 65 535: the largest windows the generic n-gram route accepts; start / end offsets close to 2^16."""
    rnd = random.Random(79)
    alpha = bytes(rnd.sample(range(1, 256), 4))
    m = 65400
    p = bytes(rnd.choices(alpha, k=m))
    t = bytearray(rnd.choices(alpha, k=3 * m))
    v = bytearray(p)
    v[100] = next(c for c in alpha if c != v[100])               # one substitution near the start
    del v[40000]                                                 # one deletion
    v.insert(65000, alpha[0])                                    # one insertion near the end
    t[70000:70000 + len(v)] = v
    t = bytes(t)
    h = engine.upload(t)
    limits = (3, 2, 2, 4)
    exp = oracle.generic_ngrams_raw(p, t, *limits)
    assert engine.generic_ngrams(h, p, *limits) == exp and len(exp) > 0
    assert max(e - s for (s, e, _d, _g) in exp) > 65000
    assert [r[:3] for r in engine.generic_ngrams_consolidated(h, p, *limits)] == oracle.consolidate(exp)
    h.release()


def test_public_api_with_a_long_pattern():
    import io
    import fuzzysearch_amd as fa
    rnd = random.Random(5)
    t = bytes(rnd.choices(b"ACGT", k=300000))
    p = bytearray(t[120000:121500])
    del p[700]
    p[20] = ord('A') if p[20] != ord('A') else ord('C')
    p = bytes(p)
    res = fa.find_near_matches(p, t, max_l_dist=2)
    assert [(x.start, x.end, x.dist) for x in res] == [(120000, 121500, 2)]
    assert bytes(res[0].matched) == t[120000:121500]
    res_f = fa.find_near_matches_in_file(p, io.BytesIO(t), max_l_dist=2)
    assert [(x.start, x.end, x.dist) for x in res_f] == [(120000, 121500, 2)]
    res_f = fa.find_near_matches_in_file(p, io.BytesIO(t), max_l_dist=2, _chunk_size=50000)
    assert [(x.start, x.end, x.dist) for x in res_f] == [(120000, 121500, 2)]


def test_stress_parity_slice():
    """20 s of benchmarks/stress_parity.py with a fixed seed (every route, random alphabets and sizes)."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "stress_parity.py"), "20", "20260926"], cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = res.stdout.decode()
    assert res.returncode == 0 and "all equal" in out, out[-3000:]
