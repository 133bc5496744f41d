"""Random GPU-vs-oracle cases that also run in a SUBPROCESS (`python -m tests.gpu_cases <what> ...`): several switches
of the library are process-wide statics read from the environment (FZ_FORCE_BIG_VERIFY, FZ_NO_SLOT_AND,
FZ_MAX_BLOCKS, FZ_NO_DIRECT), so a test that wants them set starts a fresh interpreter.  Prints "OK <n cases> <n records>"."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def edited(rnd, p, n_edits, alpha):
    v = bytearray(p)
    for _ in range(n_edits):
        q = rnd.randrange(len(v))
        op = rnd.random()
        if op < 0.4:
            v[q] = rnd.choice(alpha)
        elif op < 0.7 and len(v) > 2:
            del v[q]
        else:
            v.insert(q, rnd.choice(alpha))
    return bytes(v)


def random_cases(rnd, n_cases, ks, max_m, max_n):
    """(pattern, text, k) with planted edited copies at ragged positions (both ends included)."""
    out = []
    while len(out) < n_cases:
        alpha = bytes(rnd.sample(range(1, 256), rnd.choice([2, 3, 4, 4, 20, 200])))
        k = rnd.choice(ks)
        m = rnd.randint(k + 1, max(k + 1, min(max_m, rnd.choice([8, 24, 64, max_m]))))
        n = rnd.randint(0, max_n)
        t = bytearray(rnd.choice(alpha) for _ in range(n))
        p = bytes(rnd.choice(alpha) for _ in range(m))
        for _rep in range(3):
            if n > m + 10 and rnd.random() < 0.8:
                v = edited(rnd, p, rnd.randint(0, k), alpha)
                st = rnd.choice([0, 1, n - len(v) - 1, n - len(v), rnd.randint(0, max(0, n - len(v)))])
                st = max(0, min(st, n - len(v)))
                t[st:st + len(v)] = v
        if len(p) // (k + 1) == 0:
            continue
        out.append((p, bytes(t), k))
    return out


def run_lev_subs(engine, cases):
    import oracle
    n_rec = 0
    for (p, t, k) in cases:
        h = engine.upload(t)
        got = engine.lev_ngrams(h, p, k)
        assert got == oracle.lev_ngrams_raw(p, t, k), ("lev", p, t, k)
        n_rec += len(got)
        got = engine.subs_ngrams(h, p, k)
        assert got == oracle.subs_ngrams_raw(p, t, k), ("subs", p, t, k)
        n_rec += len(got)
        h.release()
    return n_rec


def run_pipelined(engine, cases):
    """Two searches in flight (mixed kinds), and the folded generic search, against the oracle."""
    import oracle
    n_rec = 0
    for (p, t, k) in cases:
        h = engine.upload(t)
        exp_s, exp_l = oracle.subs_ngrams_raw(p, t, k), oracle.lev_ngrams_raw(p, t, k)
        engine.subs_ngrams_begin(h, p, k)
        engine.lev_ngrams_begin(h, p, k)
        assert engine.search_end() == exp_s, ("subs, two in flight", p, t, k)
        assert engine.search_end() == exp_l, ("lev, two in flight", p, t, k)
        engine.lev_ngrams_begin(h, p, k)
        engine.lev_ngrams_begin(h, p, k)
        assert engine.search_end() == exp_l and engine.search_end() == exp_l, ("lev twice", p, t, k)
        n_rec += len(exp_s) + len(exp_l)
        if k and len(t) <= 3000 and len(p) // (k + 1) >= 1:
            try:
                want = oracle.generic_ngrams_raw(p, t, k, k, k, k)
                got = engine.generic_ngrams_consolidated(h, p, k, k, k, k)
            except NotImplementedError:
                pass
            else:
                assert [r[:3] for r in got] == oracle.consolidate(want), ("generic consolidated", p, t, k)
                engine.generic_ngrams_begin(h, p, k, k, k, k, consolidated=True)
                engine.generic_ngrams_begin(h, p, k, k, k, k)
                assert engine.search_end() == got and engine.search_end() == want, ("generic, two in flight", p, t, k)
                n_rec += len(want)
        h.release()
    return n_rec


def run_generic_windows(engine, rnd, n_cases):
    """Generic searches whose n-gram hits share windows (the window table, fz_device.h: FzGenDedup): exact and lightly
    edited copies of patterns with 2 .. 14 blocks (more hits per window than a slot lists members), copies at both ends
    of the text, dense repeats — raw stream, consolidated rows (block included: the smallest), the flag and the two-deep
    pipeline against the oracle."""
    import oracle
    n_rec = cases = 0
    while cases < n_cases:
        alpha = bytes(rnd.sample(range(1, 256), rnd.choice([3, 4, 8, 60])))
        k = rnd.choice([1, 2, 3, 4, 6, 9, 13])
        L = rnd.choice([2, 3, 4, 6])
        m = (k + 1) * L + rnd.randint(0, L - 1)
        p = bytes(rnd.choice(alpha) for _ in range(m))
        n = rnd.randint(m, 4000)
        t = bytearray(rnd.choice(alpha) for _ in range(n))
        for _rep in range(rnd.randint(1, 6)):
            v = edited(rnd, p, rnd.choice([0, 0, 0, 1, 2]), alpha)
            st = max(0, min(rnd.choice([0, n - len(v), rnd.randint(0, max(0, n - len(v)))]), n - len(v)))
            t[st:st + len(v)] = v
        if rnd.random() < 0.15:
            t = bytearray((p * (n // m + 1))[:n])                   # dense repeats: every window shared, long member lists
        t = bytes(t)
        lim = (rnd.randint(0, k), rnd.randint(0, min(k, 3)), rnd.randint(0, min(k, 3)), k)
        try:
            want = oracle.generic_ngrams_raw(p, t, *lim)
        except Exception:
            continue
        if len(want) > 200000:
            continue
        h = engine.upload(t)
        try:
            got = engine.generic_ngrams(h, p, *lim)
        except NotImplementedError:
            h.release()
            continue
        assert got == want, ("generic raw", p, t, lim)
        cons = engine.generic_ngrams_consolidated(h, p, *lim)
        assert [r[:3] for r in cons] == oracle.consolidate(want), ("generic consolidated", p, t, lim)
        from fuzzysearch_amd import _native
        assert cons == [tuple(r) for r in _native.consolidate(got)], ("consolidated rows incl. block", p, t, lim)
        assert engine.generic_ngrams_any(h, p, *lim) == (len(want) > 0)
        engine.generic_ngrams_begin(h, p, *lim)
        engine.generic_ngrams_begin(h, p, *lim, consolidated=True)
        assert engine.search_end() == want and engine.search_end() == cons, ("generic, two in flight", p, t, lim)
        h.release()
        n_rec += len(want)
        cases += 1
    return cases, n_rec


def main(argv):
    from fuzzysearch_amd import _native
    what = argv[0]
    eng = _native.Engine([0])
    rnd = random.Random(int(argv[2]) if len(argv) > 2 else 5)
    n = int(argv[1]) if len(argv) > 1 else 300
    if what == "big":
        # FZ_FORCE_BIG_VERIFY=1: every verification by fz_verify_big_kernel — small budgets (one cell per lane) up to
        # wide ones (CPL 2, 4), short and long pieces, segment ends
        cases = random_cases(rnd, n, [1, 2, 3, 5, 8, 12, 31, 32, 40, 70, 100], 260, 700)
    elif what == "slots":
        # FZ_NO_SLOT_AND=1 (the general slot form of the filter) and FZ_MAX_BLOCKS (several launches per search)
        cases = random_cases(rnd, n, [1, 2, 3, 4, 5, 7], 120, 3000)
        import numpy as np
        from tests import workloads
        seq = workloads.dna(4 << 20, 31)
        pat = workloads.dna(20, 1)
        workloads.plant_variants(seq, pat, 256, 3)
        cases.append((pat.tobytes(), seq.tobytes(), 2))
        seq = workloads.text65(2 << 20, 32)
        pat = workloads.text65(36, 2)
        workloads.plant_edits(seq, pat, 128, 4, workloads.TEXT65, lambda i: i % 4)
        cases.append((pat.tobytes(), seq.tobytes(), 3))
    elif what == "copy":
        # FZ_NO_DIRECT=1: counters and records through D2H copies (the path of searches with more records than the
        # pinned staging buffer holds) — one call at a time and two in flight, where the younger search has to wait
        # for the older one's records to leave the shared device buffer
        cases = random_cases(rnd, n, [1, 2, 3, 4, 5], 60, 3000)
        for sigma, nn, m, k in ((5, 300000, 5, 3), (4, 200000, 8, 1), (3, 100000, 12, 2)):
            alpha = bytes(rnd.sample(range(1, 256), sigma))
            cases.append((bytes(rnd.choices(alpha, k=m)), bytes(rnd.choices(alpha, k=nn)), k))   # up to 2.6e5 records per search
        n_rec = run_pipelined(eng, cases)
    elif what == "taper":
        # the scan grid's regions (FZ_TAPER_STEPS / FZ_TAPER_MIN / FZ_TAPER_WG_PER_CU): a launch big enough to taper its
        # last resident round must return the same streams however the tiles are dealt out
        import hashlib
        from tests import workloads
        seq, pat, _ = workloads.cfg2(n << 20, n)
        p = pat.tobytes()
        h = eng.upload(seq)
        dig = hashlib.sha1()
        n_rec = 0
        lev = eng.lev_ngrams(h, p, 2, as_array=True)
        for rows in (lev, eng.search_exact(h, p, as_array=True),
                     eng.subs_ngrams(h, p, 2, as_array=True), eng.lev_ngrams(h, workloads.dna(36, 9).tobytes(), 5, as_array=True)):
            dig.update(rows.tobytes())
            n_rec += len(rows)
        # ... and the END of the buffer — where the tapered workgroups of the grid's last resident round work — against the
        # ORACLE, so that whichever grid form this process runs (no regions, the default taper, a steep one) is held against
        # the reference's algorithm and not only against another run of this library: the oracle on the last 64 MiB, rows
        # that start at least 64 bytes behind the cut (their windows do not reach across it), shifted, in stream order
        import oracle
        tail = 64 << 20
        off = len(seq) - tail
        want = [(s_ + off, e_ + off, d_, g_) for (s_, e_, d_, g_) in oracle.lev_ngrams_raw(p, seq[off:].tobytes(), 2) if s_ >= 64]
        got = [tuple(int(x) for x in r) for r in lev.tolist() if int(r[0]) >= off + 64]
        assert got == want and len(want) > 30, (len(got), len(want))
        h.release()
        eng.close()
        print("OK %d %d %d" % (n, n_rec, int(dig.hexdigest()[:12], 16)))
        return
    elif what == "wf":
        # Levenshtein budgets 5 .. 15: lane-per-cell verification inside the scan kernel (default; 16 lanes per candidate up
        # to 7, 32 beyond) or in the kernel of its own (FZ_NO_WF_FUSE=1) — ragged ends, patterns up to the argument block, queues that fill up (tiny alphabets)
        cases = random_cases(rnd, n, [5, 6, 7, 8, 10, 12, 15, 20, 31], 300, 6000)   # (16 .. 31: 64 lanes, stand-alone kernel only)
        from tests import workloads
        for sigma, nn, m, k in ((2, 60000, 40, 5), (4, 400000, 30, 5), (3, 150000, 56, 7), (20, 1 << 20, 64, 6),
                                (4, 300000, 50, 8), (3, 100000, 96, 15), (20, 1 << 20, 120, 11), (4, 200000, 130, 24),
                                (2, 400000, 40, 5), (2, 300000, 75, 9), (2, 300000, 140, 19)):   # 2e4 .. 4e4 candidates: the stand-alone kernel appends its records
            alpha = bytes(rnd.sample(range(1, 256), sigma))
            pp = bytes(rnd.choices(alpha, k=m))
            tt = bytearray(rnd.choices(alpha, k=nn))
            for _ in range(40):
                v = edited(rnd, pp, rnd.randint(0, k), alpha)
                st = rnd.randint(0, nn - len(v))
                tt[st:st + len(v)] = v
            cases.append((pp, bytes(tt), k))
        seq = workloads.text65(4 << 20, 41)
        pat = workloads.text65(64, 3)
        workloads.plant_edits(seq, pat, 200, 5, workloads.TEXT65, lambda i: i % 6)
        cases.append((pat.tobytes(), seq.tobytes(), 5))
        n_rec = 0
        import oracle
        for (pp, tt, k) in cases:                               # (Levenshtein only: the substitutions form has no such path)
            h = eng.upload(tt)
            got = eng.lev_ngrams(h, pp, k)
            assert got == oracle.lev_ngrams_raw(pp, tt, k), ("lev", pp, tt[:200], k)
            n_rec += len(got)
            h.release()
        eng.close()
        print("OK %d %d" % (len(cases), n_rec))
        return
    elif what == "windows":
        # the generic search's window table, on (default) and off (FZ_GEN_NO_DEDUP=1: every hit runs the automaton)
        n_cases, n_rec = run_generic_windows(eng, rnd, n)
        eng.close()
        print("OK %d %d" % (n_cases, n_rec))
        return
    else:
        raise SystemExit("unknown case set %r" % what)
    n_rec = n_rec + run_lev_subs(eng, cases) if what == "copy" else run_lev_subs(eng, cases)
    eng.close()
    print("OK %d %d" % (len(cases), n_rec))


if __name__ == "__main__":
    main(sys.argv[1:])
