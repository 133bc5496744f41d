"""The FZ_HD verification functions of fuzzysearch_amd/csrc/fz_device.h — the exact code the GPU
kernels run per candidate (banded ring-buffer expansion, window clamps, shard geometry) — compiled
with g++ (tests/host_emul.cpp) and checked lane by lane against the oracle.  CPU only."""
import ctypes
import os
import random
import subprocess
import tempfile

import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


class OutRec(ctypes.Structure):
    _fields_ = [("start", ctypes.c_int64), ("end", ctypes.c_int64), ("dist", ctypes.c_int32), ("block", ctypes.c_int32)]


@pytest.fixture(scope="module")
def emul():
    out = os.path.join(tempfile.gettempdir(), "fz_hostemul_%d.so" % os.getpid())
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall",
                           os.path.join(HERE, "host_emul.cpp"), "-o", out])
    L = ctypes.CDLL(out)
    L.emul_search.restype = ctypes.c_int64
    L.emul_search.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64,
                              ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                              ctypes.POINTER(OutRec), ctypes.c_int64]
    L.emul_expand.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32,
                              ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.emul_wf_expand.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32,
                                 ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                 ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.emul_expand_bits.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32,
                                   ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.emul_search_bits.restype = ctypes.c_int64
    L.emul_search_bits.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32,
                                   ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                   ctypes.POINTER(OutRec), ctypes.c_int64]
    yield L
    os.remove(out)


def _search(L, mode, p, t, k, buf_off=0, buf_len=None, own_lo=0, own_hi=None):
    n = len(t)
    cap = 8192
    out = (OutRec * cap)()
    c = L.emul_search(mode, p, len(p), t, n, k, buf_off, n - buf_off if buf_len is None else buf_len,
                      own_lo, n if own_hi is None else own_hi, out, cap)
    assert 0 <= c <= cap
    return [(out[i].start, out[i].end, out[i].dist, out[i].block) for i in range(c)]


def _case(rnd, max_n=80, max_m=24, max_k=5):
    sigma = rnd.choice([2, 2, 3, 4])
    alpha = bytes(rnd.sample(range(65, 91), sigma))
    n = rnd.randint(0, max_n)
    t = bytes(rnd.choice(alpha) for _ in range(n))
    k = rnd.randint(1, max_k)
    m = rnd.randint(k + 1, max_m)
    if rnd.random() < 0.6 and n >= m:
        st = rnd.randint(0, n - m)
        p = bytearray(t[st:st + m])
        for _ in range(rnd.randint(0, k)):
            q = rnd.randrange(len(p))
            op = rnd.random()
            if op < 0.4:
                p[q] = rnd.choice(alpha)
            elif op < 0.7 and len(p) > k + 1:
                del p[q]
            else:
                p.insert(q, rnd.choice(alpha))
        p = bytes(p)
    else:
        p = bytes(rnd.choice(alpha) for _ in range(m))
    return p, t, k


def test_banded_ring_expand_equals_full_dp(emul):
    rnd = random.Random(5)
    for _ in range(60000):
        alpha = bytes(rnd.sample(range(65, 91), rnd.choice([2, 3, 4])))
        sub = bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, 20)))
        b = rnd.randint(0, 7)
        win = bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, len(sub) + b + 3)))
        if rnd.random() < 0.6 and sub:
            w = bytearray(sub)
            for _ in range(rnd.randint(0, b + 1)):
                q = rnd.randrange(len(w) + 1)
                op = rnd.random()
                if op < 0.4 and q < len(w):
                    w[q] = rnd.choice(alpha)
                elif op < 0.7 and q < len(w):
                    del w[q]
                else:
                    w.insert(q, rnd.choice(alpha))
            win = bytes(w) + bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, 3)))
        d, c = ctypes.c_uint32(), ctypes.c_uint32()
        r = emul.emul_expand(sub, len(sub), win, len(win), b, ctypes.byref(d), ctypes.byref(c))
        assert ((d.value, c.value) if r else (None, None)) == oracle.expand(sub, win, b), (sub, win, b)


def _edited(rnd, sub, alpha, n_edits, tail):
    w = bytearray(sub)
    for _ in range(n_edits):
        q = rnd.randrange(len(w) + 1)
        op = rnd.random()
        if op < 0.4 and q < len(w):
            w[q] = rnd.choice(alpha)
        elif op < 0.7 and q < len(w):
            del w[q]
        else:
            w.insert(q, rnd.choice(alpha))
    return bytes(w) + bytes(rnd.choice(alpha) for _ in range(tail))


def test_bit_vector_expand_equals_full_dp(emul):
    """fz_expand_bits / fz_bits_column (the column recurrence the fused bit-vector verification runs per lane) against the
    oracle's full DP with the LAST arg-min rule, on >= 1e5 random pieces: one-word pieces up to 64 rows, two-word pieces
    up to 128, small alphabets (ties along the bottom row are the rule there), windows shorter and longer than the piece,
    empty pieces and windows, budgets from 0 to beyond the piece."""
    rnd = random.Random(61)
    done = {1: 0, 2: 0, 4: 0}
    for it in range(150000):
        nw = 1 if it < 100000 else 2 if it < 125000 else 4          # 4 names the 32-bit form
        alpha = bytes(rnd.sample(range(1, 256), rnd.choice([1, 2, 2, 3, 4, 4, 20])))
        top = 32 if nw == 4 else 64 * nw
        ln = rnd.choice([0, 1, 2, top - 1, top, rnd.randint(0, top), rnd.randint(0, 24), rnd.randint(0, 24)])
        sub = bytes(rnd.choice(alpha) for _ in range(ln))
        b = rnd.choice([0, 1, 2, 3, 5, 8, 13, 21, ln, ln + 1])
        if rnd.random() < 0.6 and sub:
            win = _edited(rnd, sub, alpha, rnd.randint(0, b + 1), rnd.randint(0, 3))
            win = win[:rnd.choice([len(win), len(win), rnd.randint(0, len(win))])]
        else:
            win = bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, min(250, ln + b + 3))))
        win = win[:250]
        d, c = ctypes.c_uint32(), ctypes.c_uint32()
        r = emul.emul_expand_bits(nw, sub, len(sub), win, len(win), b, ctypes.byref(d), ctypes.byref(c))
        assert r in (0, 1)
        assert ((d.value, c.value) if r else (None, None)) == oracle.expand(sub, win, b), (nw, sub, win, b)
        done[nw] += 1
    assert done[1] >= 100000 and done[2] >= 25000 and done[4] >= 25000


def _search_bits(L, nw, p, t, k, buf_off=0, buf_len=None, own_lo=0, own_hi=None):
    n = len(t)
    cap = 1 << 15
    out = (OutRec * cap)()
    c = L.emul_search_bits(nw, p, len(p), t, n, k, buf_off, n - buf_off if buf_len is None else buf_len,
                           own_lo, n if own_hi is None else own_hi, out, cap)
    assert 0 <= c <= cap, c
    return [(out[i].start, out[i].end, out[i].dist, out[i].block) for i in range(c)]


def test_bit_vector_per_hit_verification_equals_oracle(emul):
    """fz_verify_lev_bits — both expansions of a hit in one loop, both pieces taken out of the two whole-pattern tables by
    their top-bit masks — against the oracle's raw stream: patterns up to 64 (one word) and 128 (two words) characters,
    budgets 1 .. 21, hits at both sequence ends (Python-slice clamps), planted occurrences with up to k edits, and the
    shard geometry with a poisoned halo."""
    rnd = random.Random(62)
    done = 0
    while done < 7500:
        nw = 1 if done < 4500 else 2 if done < 6000 else 4
        top = 32 if nw == 4 else 64 * nw
        sigma = rnd.choice([2, 3, 4, 4, 4, 20])
        alpha = bytes(rnd.sample(range(1, 256), sigma))
        k = rnd.choice([1, 2, 3, 4, 5, 6, 8, 8, 12, 21])
        lo_m = max(k + 1, 2)
        m = rnd.choice([lo_m, top, top - 1, rnd.randint(lo_m, max(lo_m, top)), rnd.randint(lo_m, max(lo_m, 30))])
        if m > top or m // (k + 1) == 0:
            continue
        p = bytes(rnd.choice(alpha) for _ in range(m))
        n = rnd.choice([0, rnd.randint(0, m), rnd.randint(m, 3 * m + 40), rnd.randint(m, 400)])
        t = bytearray(rnd.choice(alpha) for _ in range(n))
        for _ in range(rnd.randint(0, 3)):
            v = _edited(rnd, p, alpha, rnd.randint(0, k), 0)
            if len(v) <= n:
                at = rnd.choice([0, n - len(v), rnd.randint(0, n - len(v))])
                t[at:at + len(v)] = v
        t = bytes(t)
        want = oracle.lev_ngrams_raw(p, t, k)
        if len(want) > 20000:
            continue
        assert _search_bits(emul, nw, p, t, k) == want, (nw, p, t, k)
        if n >= 3 and done % 3 == 0:
            halo = m + k
            cut = rnd.randint(0, n)
            a = _search_bits(emul, nw, p, t, k, 0, min(n, cut + halo), 0, cut)
            lo = max(0, cut - halo)
            b = _search_bits(emul, nw, p, t, k, lo, n - lo, cut, n)
            assert sorted(a + b, key=lambda r: r[3]) == want, (nw, p, t, k, cut)
        done += 1


def test_per_hit_verification_equals_oracle(emul):
    rnd = random.Random(6)
    for _ in range(8000):
        p, t, k = _case(rnd)
        if len(p) // (k + 1) == 0:
            continue
        assert _search(emul, 1, p, t, k) == oracle.lev_ngrams_raw(p, t, k), (p, t, k)
        assert _search(emul, 2, p, t, k) == oracle.subs_ngrams_raw(p, t, k), (p, t, k)


def test_shard_geometry_with_poisoned_halo(emul):
    """Two shards with exactly (m + k) halo bytes; everything outside a shard buffer is poison in the
    emulator, so any read beyond the halo would corrupt the result."""
    rnd = random.Random(7)
    for _ in range(6000):
        p, t, k = _case(rnd)
        n = len(t)
        if len(p) // (k + 1) == 0 or n < 3:
            continue
        halo = len(p) + k
        cut = rnd.randint(0, n)
        a = _search(emul, 1, p, t, k, 0, min(n, cut + halo), 0, cut)
        lo = max(0, cut - halo)
        b = _search(emul, 1, p, t, k, lo, n - lo, cut, n)
        merged = sorted(a + b, key=lambda r: r[3])       # stable: block-major, shards keep idx order
        assert merged == oracle.lev_ngrams_raw(p, t, k), (p, t, k, cut)


def test_generic_automaton_struct_and_packed_steps_equal_oracle(emul):
    """fz_generic_step (the host-tested statement of generic_search.py:57-177) and fz_generic_step_packed (what
    fz_lp_kernel runs per candidate) driven over whole sequences: both emit the oracle's list, and the two forms
    agree candidate by candidate."""
    sig = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
           ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(OutRec), ctypes.c_int64]
    for fn in (emul.emul_generic_lp, emul.emul_generic_lp_packed):
        fn.restype = ctypes.c_int64
        fn.argtypes = sig
    rnd = random.Random(11)
    cap = 1 << 16
    out = (OutRec * cap)()
    done = 0
    while done < 2500:
        p, t, k = _case(rnd, max_n=60, max_m=16, max_k=4)
        limits = (rnd.randint(0, k), rnd.randint(0, k), rnd.randint(0, k))
        max_l = min(k, sum(limits))
        if len(p) > 60000 or max_l >= len(p):
            continue
        want = [r[:3] for r in oracle.generic_lp_raw(p, t, limits[0], limits[1], limits[2], max_l)]
        for fn in (emul.emul_generic_lp, emul.emul_generic_lp_packed):
            c = fn(p, len(p), t, len(t), limits[0], limits[1], limits[2], max_l, out, cap)
            assert 0 <= c <= cap, (fn, p, t, limits, max_l)
            assert [(out[i].start, out[i].end, out[i].dist) for i in range(c)] == want, (p, t, limits, max_l)
        done += 1
    # round 5: long patterns (the 64-bit equality words of fz_generic_step_bits up to their last bit, the sentinel at m = 64)
    # and larger deletion budgets (find-first-set beyond the first two skip positions); emul_generic_lp_packed compares the
    # bit-parallel step with the packed one on every candidate
    done = 0
    while done < 300:
        sigma = rnd.choice([2, 3, 4, 20])
        alpha = bytes(rnd.sample(range(1, 256), sigma))
        m = rnd.choice([31, 32, 33, 48, 63, 64, 64, rnd.randint(17, 64)])
        k = rnd.randint(1, 6)
        p = bytes(rnd.choice(alpha) for _ in range(m))
        t = bytearray(rnd.choice(alpha) for _ in range(rnd.randint(m, 260)))
        v = bytearray(p)
        for _ in range(rnd.randint(0, k)):
            q = rnd.randrange(len(v))
            r = rnd.random()
            if r < 0.34:
                v[q] = rnd.choice(alpha)
            elif r < 0.67 and len(v) > 2:
                del v[q]
            else:
                v.insert(q, rnd.choice(alpha))
        at = rnd.randint(0, max(0, len(t) - len(v)))
        t[at:at + len(v)] = v
        t = bytes(t)
        limits = (rnd.randint(0, k), rnd.randint(0, k), rnd.randint(0, k))
        max_l = min(k, sum(limits))
        if max_l == 0:
            continue
        want = [r[:3] for r in oracle.generic_lp_raw(p, t, limits[0], limits[1], limits[2], max_l)]
        c = emul.emul_generic_lp_packed(p, len(p), t, len(t), limits[0], limits[1], limits[2], max_l, out, cap)
        assert 0 <= c <= cap, (p, t, limits, max_l, c)
        assert [(out[i].start, out[i].end, out[i].dist) for i in range(c)] == want, (p, t, limits, max_l)
        done += 1


def test_generic_ngram_search_as_ordered_on_the_device_equals_oracle(emul):
    """The whole generic n-gram pipeline in the form the GPU runs it with device-side ordering — hits in a scrambled
    order, packed automaton per hit window, row counts, first rows by key, scatter through fz_gen_row — emits the
    oracle's list (blocks in order, hits by index, matches in emission order, duplicates included).  Round 4: with the
    window table (the scan enters every hit; only the smallest block of a window runs, the others take its rows; member
    lists shortened to 2 so that windows with more hits than a slot lists occur) and with the starts of a window dealt
    out to 2 or 4 waves whose sorted match buffers are merged by rank (fz_gen_hit_kernel).  Round 5 (mode bit 3): the window
    walked in the kernel's two runs (spawning characters up to the closed-form n_spawn, then until nothing is alive) with
    the bit-parallel step on the window's equality words."""
    fn = emul.emul_generic_ngrams_ordered
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                   ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(OutRec), ctypes.c_int64]
    rnd = random.Random(12)
    cap = 1 << 17
    out = (OutRec * cap)()
    done = total = 0
    modes = [0, 1, 1 | (1 << 1), 1 | (2 << 1), (1 << 1), 1 | (1 << 1) | (2 << 8), 1 | (2 << 8), 1 | 8, 8 | (1 << 1), 1 | 8 | (2 << 1)]
    while done < 1500:
        p, t, k = _case(rnd, max_n=120, max_m=24, max_k=4)
        if done % 3 == 0 and len(t) > 3 * len(p):                   # exact copies: every block hits, the windows are shared
            at = rnd.randint(0, len(t) - len(p))
            t = t[:at] + p + t[at + len(p):]
        limits = (rnd.randint(0, k), rnd.randint(0, k), rnd.randint(0, k))
        max_l = min(k, sum(limits))
        if max_l == 0 or len(p) // (max_l + 1) == 0:
            continue
        want = oracle.generic_ngrams_raw(p, t, limits[0], limits[1], limits[2], max_l)
        for mode in (modes if done % 4 == 0 else [modes[done % len(modes)]]):
            c = fn(p, len(p), t, len(t), limits[0], limits[1], limits[2], max_l, done, mode, out, cap)
            assert 0 <= c <= cap, (c, mode, p, t, limits, max_l)
            got = [(out[i].start, out[i].end, out[i].dist, out[i].block) for i in range(c)]
            assert got == [tuple(r) for r in want], (mode, p, t, limits, max_l)
        done += 1
        total += c
    assert total > 20000


def test_generic_search_consolidated_on_the_device_equals_oracle(emul):
    """fz_generic_ngrams_consolidated as the GPU runs it: every running hit folds its matches into (hull, best) pairs in
    slices of 64 lanes (fz_gen_hit_kernel's fold, ballots restated as loops), a pair with an empty hull is left once per
    member of its window (zero-length matches never merge: the reference keeps one per n-gram hit), and the pairs go
    through the second stage — against consolidate_overlapping_matches (common.py:185-189) of the oracle's raw stream.
    With and without the window table, 1 / 2 / 4 waves per hit, short member lists, the kernel's two-run window walk."""
    fn = emul.emul_generic_ngrams_ordered
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                   ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(OutRec), ctypes.c_int64]
    rnd = random.Random(31)
    cap = 1 << 16
    out = (OutRec * cap)()
    done = total = zero_len = 0
    modes = [16, 16 | 1, 16 | 1 | 8, 16 | 1 | (1 << 1), 16 | (2 << 1), 16 | 1 | (2 << 1) | 8, 16 | 1 | (2 << 8), 16 | 1 | (1 << 1) | (2 << 8)]
    while done < 1500:
        p, t, k = _case(rnd, max_n=120, max_m=24, max_k=4)
        if done % 3 == 0 and len(t) > 3 * len(p):                   # exact copies: every block hits, the windows are shared
            at = rnd.randint(0, len(t) - len(p))
            t = t[:at] + p + t[at + len(p):]
        limits = (rnd.randint(0, k), rnd.randint(0, k), rnd.randint(0, k))
        max_l = min(k, sum(limits))
        if max_l == 0 or len(p) // (max_l + 1) == 0:
            continue
        raw = oracle.generic_ngrams_raw(p, t, limits[0], limits[1], limits[2], max_l)
        want = [tuple(r) for r in oracle.consolidate([tuple(r)[:3] for r in raw])]
        zero_len += sum(1 for r in want if r[0] == r[1])
        for mode in (modes if done % 4 == 0 else [modes[done % len(modes)]]):
            c = fn(p, len(p), t, len(t), limits[0], limits[1], limits[2], max_l, done, mode, out, cap)
            assert 0 <= c <= cap, (c, mode, p, t, limits, max_l)
            got = [(out[i].start, out[i].end, out[i].dist) for i in range(c)]
            assert got == want, (mode, p, t, limits, max_l)
        done += 1
        total += c
    assert total > 1500 and zero_len > 20


def test_levenshtein_lp_struct_and_slot_steps_equal_oracle(emul):
    """fz_levlp_step (the statement of levenshtein.py:52-148) and fz_levlp_step_slots (what fz_lp_kernel stores from since
    round 4: no arrays, no scratch memory) driven over whole sequences: the oracle's list, and the two forms agree
    candidate by candidate."""
    fn = emul.emul_lev_lp
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(OutRec), ctypes.c_int64]
    rnd = random.Random(14)
    cap = 1 << 16
    out = (OutRec * cap)()
    done = nonempty = 0
    while done < 3000:
        p, t, k = _case(rnd, max_n=70, max_m=12, max_k=4)
        if k >= len(p):
            continue
        want = [r[:3] for r in oracle.lev_lp_raw(p, t, k)]
        c = fn(p, len(p), t, len(t), k, out, cap)
        assert 0 <= c <= cap, (c, p, t, k)
        assert [(out[i].start, out[i].end, out[i].dist) for i in range(c)] == want, (p, t, k)
        done += 1
        nonempty += bool(want)
    assert nonempty > 1500


def test_lane_per_cell_expansion_equals_oracle(emul):
    """The wavefront form of the verification (fz_wf_rows / fz_wf_pick: one lane per band cell, 16 / 32 / 64 lanes per
    candidate) restated lane by lane on the host: band K >= budget, windows shorter and longer than the pattern piece,
    the early exit — against the reference's _expand (levenshtein_ngram.py:8-19 through the oracle).  And the property the
    device comment states: rows run with a LARGER window and budget give the same pick once the bottom row is read with
    the smaller ones (a cell only depends on cells of smaller or equal columns)."""
    rnd = random.Random(41)
    d, c = ctypes.c_uint32(), ctypes.c_uint32()
    n_ok = 0
    for it in range(12000):
        sigma = rnd.choice([2, 2, 3, 4, 20])
        alpha = bytes(rnd.sample(range(65, 91), sigma))
        gw = rnd.choice([16, 16, 32, 64])
        K = rnd.randint(1, (gw - 1) // 2)
        budget = rnd.randint(0, K)
        sub = bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, 70)))
        if rnd.random() < 0.7:                                   # a lightly edited copy, then some tail
            w = bytearray(sub)
            for _ in range(rnd.randint(0, budget + 1)):
                q = rnd.randrange(len(w) + 1)
                op = rnd.random()
                if op < 0.4 and q < len(w):
                    w[q] = rnd.choice(alpha)
                elif op < 0.7 and q < len(w):
                    del w[q]
                else:
                    w.insert(q, rnd.choice(alpha))
            win = bytes(w) + bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, 4)))
        else:
            win = bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, 80)))
        win = win[:len(sub) + budget] if rnd.random() < 0.8 else win     # (the search never offers more than sublen + budget)
        want = oracle.expand(sub, win, budget)
        r = emul.emul_wf_expand(gw, K, sub, len(sub), win, len(win), budget, len(win), budget, ctypes.byref(d), ctypes.byref(c))
        assert r >= 0
        assert ((d.value, c.value) if r else (None, None)) == want, (gw, K, budget, sub, win)
        n_ok += r
        # rows with the full band budget and a longer window, the pick with the narrower ones
        longer = win + bytes(rnd.choice(alpha) for _ in range(rnd.randint(0, K - budget + 2)))
        r2 = emul.emul_wf_expand(gw, K, sub, len(sub), longer, len(longer), K, len(win), budget, ctypes.byref(d), ctypes.byref(c))
        assert ((d.value, c.value) if r2 else (None, None)) == want, ("narrowed", gw, K, budget, sub, win, longer)
    assert n_ok > 3000
