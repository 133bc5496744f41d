"""-m gpu: the fused bit-vector verification (fz_scan_kernel<..., 1 / 2>: fz_verify_lev_bits on 64 / 128-bit columns, one
candidate per lane, inside the scan) against the oracle — ordered, bit-exact raw streams through the C-ABI.

What the form has to get right (_levenshtein_ngrams.pyx:9-154, levenshtein_ngram.py:177-191): the
prefix-distance boundary D[0][j] = j, the LAST arg-min over the bottom row from the column-0 baseline, the left budget
k - dR, Python-slice clamps at both sequence ends; and its own machinery: both pieces of a hit out of two whole-pattern
tables, full 64-candidate passes with the rest of the queue moved to its front, tiles taken in several block-range passes
where the data is denser than the queue, launches of up to 16 blocks."""
import random

import numpy as np
import pytest

import oracle
from fuzzysearch_amd import _native
from tests import workloads

pytestmark = pytest.mark.gpu

BITS = (_native.FORM_FUSED_BITS1, _native.FORM_FUSED_BITS2, _native.FORM_FUSED_BITS32)


def _form_of(m):
    return BITS[2] if m <= 32 else BITS[0] if m <= 64 else BITS[1]


def _reload_switches():
    _native.load_library().fz_debug_reload_switches()


@pytest.fixture
def all_budgets(monkeypatch):
    """Every Levenshtein budget through the bit-vector form (the default keeps the register band for the smallest)."""
    monkeypatch.setenv("FZ_BITS_MIN_K", "1")
    _reload_switches()
    yield
    monkeypatch.delenv("FZ_BITS_MIN_K")
    _reload_switches()


def _edited(rnd, p, alpha, n_edits):
    v = bytearray(p)
    for _ in range(n_edits):
        q = rnd.randrange(len(v) + 1)
        op = rnd.random()
        if op < 0.4 and q < len(v):
            v[q] = rnd.choice(alpha)
        elif op < 0.7 and q < len(v) and len(v) > 1:
            del v[q]
        else:
            v.insert(q, rnd.choice(alpha))
    return bytes(v)


def _case(rnd, ks, max_m, max_n=600):
    sigma = rnd.choice([2, 3, 4, 4, 4, 20])
    alpha = bytes(rnd.sample(range(1, 256), sigma))
    k = rnd.choice(ks)
    lo_m = k + 1
    m = rnd.choice([lo_m, max_m, max_m - 1, 64, 65, rnd.randint(lo_m, max(lo_m, max_m)), rnd.randint(lo_m, max(lo_m, 3 * (k + 1)))])
    if m < lo_m or m > max_m:
        m = rnd.randint(lo_m, max(lo_m, max_m))
    p = bytes(rnd.choice(alpha) for _ in range(m))
    n = rnd.choice([0, rnd.randint(0, m), rnd.randint(m, 2 * m + 40), rnd.randint(m, max_n)])
    t = bytearray(rnd.choice(alpha) for _ in range(n))
    for _ in range(rnd.randint(0, 3)):
        v = _edited(rnd, p, alpha, rnd.randint(0, k))
        if len(v) <= n:
            at = rnd.choice([0, n - len(v), rnd.randint(0, n - len(v))])
            t[at:at + len(v)] = v
    return p, bytes(t), k


def test_random_and_planted_small_cases_every_budget(engine, all_budgets):
    """Budgets 1 .. 31, patterns up to 128 characters (one- and two-word columns), sequences from empty to a few hundred
    bytes with occurrences planted at both ends: the raw stream equals the oracle's and the form that ran is the bit-vector one."""
    rnd = random.Random(601)
    forms = set()
    done = 0
    while done < 1500:
        p, t, k = _case(rnd, list(range(1, 32)), 128)
        if len(p) // (k + 1) == 0:
            continue
        want = oracle.lev_ngrams_raw(p, t, k)
        if len(want) > 30000:
            continue
        h = engine.upload(t)
        got = engine.lev_ngrams(h, p, k)
        st = engine.stats()
        h.release()
        assert got == want, (p, t, k)
        if len(t):
            assert st["verify_form"] == _form_of(len(p)), (len(p), k, st)
            forms.add(st["verify_form"])
        done += 1
    assert forms == set(BITS)


def test_default_routing_budgets_5_to_31(engine):
    """Without any switch: Levenshtein budgets from 5 on and patterns up to 128 characters take the bit-vector form, whatever
    an earlier search of the context saw (round 5 chose the form of budgets 8 .. 15 by the previous call's candidate density)."""
    rnd = random.Random(602)
    t = workloads.dna(1 << 20, 5).tobytes()
    h = engine.upload(t)
    # (9, 2) and (8, 1): budgets below 3 whose pattern lets expect dense candidates (3- / 4-character n-grams over 4 letters)
    for m, k in [(54, 8), (30, 5), (64, 5), (64, 12), (100, 20), (128, 31), (65, 6), (40, 9), (9, 2), (8, 1), (20, 3)]:
        p = workloads.dna(m, 100 + m + k).tobytes()
        tt = bytearray(t)
        at = rnd.randrange(1000, len(t) - 1000)
        v = _edited(rnd, p, b"ACGT", k)
        tt[at:at + len(v)] = v
        h2 = engine.upload(bytes(tt))
        got = engine.lev_ngrams(h2, p, k)
        st = engine.stats()
        h2.release()
        assert st["verify_form"] == _form_of(m), (m, k, st)
        assert got == oracle.lev_ngrams_raw(p, bytes(tt), k), (m, k)
        assert len(got) >= 1
    h.release()


@pytest.mark.parametrize("m,k,mib", [(54, 8, 32), (20, 4, 8), (20, 3, 16), (32, 7, 8), (100, 20, 2), (128, 15, 8), (12, 3, 2)])
def test_dense_candidates_full_passes_and_block_ranges(engine, all_budgets, m, k, mib):
    """DNA, where n-gram hits come by the hundred per tile: queues that fill up (full passes, remainders moved to the front),
    tiles denser than the queue (several block-range passes per tile), patterns of 9 .. 16 blocks in one launch and of more
    than 16 in several — with planted occurrences (0 .. k edits) all over the sequence."""
    n = mib << 20
    seq = workloads.dna(n, 700 + m)
    p = workloads.dna(m, 710 + m)
    workloads.plant_edits(seq, p, 200, 720 + m, workloads.DNA, lambda i: i % (k + 1))
    t = seq.tobytes()
    h = engine.upload(seq)
    got = engine.lev_ngrams(h, p.tobytes(), k)
    st = engine.stats()
    h.release()
    assert st["verify_form"] in BITS
    want = oracle.lev_ngrams_raw(p.tobytes(), t, k)
    assert len(want) >= 150
    assert got == want


def test_queue_sizes_and_degenerate_density(engine, all_budgets, monkeypatch):
    """The queue's capacity only steers: with the smallest (FZ_BITS_QCAP=64) and a large one the streams are the oracle's — on
    DNA with four-character n-grams (every tile overflows the small queue: block-range passes down to single blocks) and on
    a run of one character met by an n-gram of that character (a tile overflows the EMPTY queue with one block: enumerated)."""
    rng = np.random.default_rng(9)
    seq = workloads.dna(1 << 20, 41)
    p = workloads.dna(24, 42)                                   # k = 5: L = 4, 6 blocks
    workloads.plant_edits(seq, p, 50, 43, workloads.DNA, lambda i: i % 6)
    t = seq.tobytes()
    runs = bytearray(workloads.dna(1 << 18, 44).tobytes())
    runs[50000:58000] = b"A" * 8000
    runs[200000:200040] = b"A" * 40
    pa = b"AAAAAACGTACGTTGCAAAAA"                               # k = 2: L = 7; k = 4: L = 4 — blocks of A's inside runs of A's
    want1 = oracle.lev_ngrams_raw(p.tobytes(), t, 5)
    for qcap in ("64", "192"):
        monkeypatch.setenv("FZ_BITS_QCAP", qcap)
        _reload_switches()
        h = engine.upload(seq)
        assert engine.lev_ngrams(h, p.tobytes(), 5) == want1
        assert engine.stats()["verify_form"] == BITS[2]
        h.release()
        h = engine.upload(bytes(runs))
        for k in (2, 4):
            assert engine.lev_ngrams(h, pa, k) == oracle.lev_ngrams_raw(pa, bytes(runs), k), (qcap, k)
        h.release()
    monkeypatch.delenv("FZ_BITS_QCAP")
    _reload_switches()
    assert rng is not None


def test_old_forms_stay_reachable_and_equal(engine, monkeypatch):
    """FZ_NO_BITS: budgets 5 .. 15 through round 5's lane-per-cell forms — the same streams (the A/B the bench's cliff map and
    profiles/r06_verify_regimes.txt rest on)."""
    seq = workloads.dna(4 << 20, 51)
    p = workloads.dna(54, 7).tobytes()
    workloads.plant_edits(seq, np.frombuffer(p, dtype=np.uint8), 40, 52, workloads.DNA, lambda i: i % 9)
    h = engine.upload(seq)
    new = engine.lev_ngrams(h, p, 8)
    assert engine.stats()["verify_form"] == BITS[0]
    monkeypatch.setenv("FZ_NO_BITS", "1")
    _reload_switches()
    old = engine.lev_ngrams(h, p, 8)
    assert engine.stats()["verify_form"] in (_native.FORM_FUSED_CELLS, _native.FORM_KERNEL)
    monkeypatch.delenv("FZ_NO_BITS")
    _reload_switches()
    h.release()
    assert new == old and len(new) >= 30


@pytest.mark.parametrize("m,k,mib", [(20, 4, 4), (12, 2, 4), (12, 3, 2), (9, 2, 1), (32, 7, 2)])
def test_substitutions_only_with_dense_candidates(engine, m, k, mib):
    """Substitutions-only n-gram searches of short DNA patterns (4- and 3-character n-grams: 50 .. 800 candidates per tile and
    wave) run the Hamming count under the bit-vector forms' queue discipline (fz_scan_kernel<..., 3>): the oracle's stream,
    duplicates across blocks and order included; the group-best form and has_near_match on top."""
    seq = workloads.dna(mib << 20, 800 + m)
    p = workloads.dna(m, 810 + m)
    workloads.plant_edits(seq, p, 100, 820 + m, workloads.DNA, lambda i: i % (k + 1), kinds=(1,))
    t = seq.tobytes()
    h = engine.upload(seq)
    got = engine.subs_ngrams(h, p.tobytes(), k)
    want = oracle.subs_ngrams_raw(p.tobytes(), t, k)
    assert len(want) >= 100
    assert got == want
    assert engine.subs_ngrams_any(h, p.tobytes(), k) is True
    h.release()


def test_sharded_sequence_two_device_states(all_budgets):
    """The bit-vector form on a sharded sequence (two device states on one GPU: contiguous shards, (m + k)-byte halos, hits owned
    by index, global clamps): the merged stream is the oracle's, for hits on both sides of the shard boundary."""
    eng = _native.Engine([0, 0])
    try:
        rnd = random.Random(77)
        for m, k in [(54, 8), (20, 3), (100, 20)]:
            seq = workloads.dna(6 << 20, 900 + m)
            p = workloads.dna(m, 910 + m)
            workloads.plant_edits(seq, p, 60, 920 + m, workloads.DNA, lambda i: i % (k + 1))
            mid = len(seq) // 2
            for delta in (-m - k, -m // 2, -3, 0, 5):                  # occurrences across and next to the shard boundary
                v = _edited(rnd, p.tobytes(), b"ACGT", rnd.randint(0, k))
                seq[mid + delta:mid + delta + len(v)] = np.frombuffer(v, dtype=np.uint8)
            h = eng.upload(seq)
            got = eng.lev_ngrams(h, p.tobytes(), k)
            assert eng.stats()["verify_form"] in BITS
            h.release()
            assert got == oracle.lev_ngrams_raw(p.tobytes(), seq.tobytes(), k), (m, k)
    finally:
        eng.close()
