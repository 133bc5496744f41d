"""-m gpu: the RCCL exchange behind the C-ABI (fz_comm_*, no torch) — as far as ONE GPU allows: a communicator of
one rank (ncclCommInitRank / ncclCommInitAll with world 1) drives the whole collective code path: device-side
snapshots of counters + records, ncclAllGather on the communicator's own stream, capacity that follows the counts,
the two-deep pipeline with the gather of search i next to the scan of search i + 1.  (More than one rank needs more
than one GPU: RCCL refuses two ranks on one device; the N > 1 data path — shards, halos, ownership, merge order —
is covered by tests/test_gpu_multi_device.py and the gloo tests.)"""
import numpy as np
import pytest

import oracle
from tests import workloads

pytestmark = pytest.mark.gpu


def _rows(arr):
    return [tuple(int(x) for x in r) for r in arr.tolist()]


@pytest.fixture()
def comm_engine():
    from fuzzysearch_amd import _native
    eng = _native.Engine([0])
    eng.comm_init_rank(eng.comm_unique_id(), 1, 0)
    yield eng
    eng.close()


def test_collective_search_of_one_rank_equals_the_oracle(comm_engine):
    eng = comm_engine
    assert eng.comm_info() == (1, 0, True)
    seq = workloads.dna(8 << 20, 77)
    pattern = workloads.dna(20, 1)
    workloads.plant_variants(seq, pattern, 512, 5)
    p, t = pattern.tobytes(), seq.tobytes()
    h = eng.upload(seq)
    exp = oracle.lev_ngrams_raw(p, t, 2)
    assert _rows(eng.lev_ngrams(h, p, 2, as_array=True)) == exp
    assert eng.stats()["raw_matches"] == len(exp) and len(exp) > 500
    # two searches in flight, different patterns, gathered in launch order
    p2 = workloads.dna(24, 9).tobytes()
    exp2 = oracle.lev_ngrams_raw(p2, t, 3)
    eng.lev_ngrams_begin(h, p, 2)
    eng.lev_ngrams_begin(h, p2, 3)
    assert _rows(eng.lev_ngrams_end(as_array=True)) == exp
    eng.lev_ngrams_begin(h, p, 2)
    assert _rows(eng.lev_ngrams_end(as_array=True)) == exp2
    assert _rows(eng.lev_ngrams_end(as_array=True)) == exp
    # wide budget: the slot-per-hit wavefront kernel's records (with empty slots) travel the same way
    p3 = workloads.dna(48, 11).tobytes()
    assert _rows(eng.lev_ngrams(h, p3, 6, as_array=True)) == oracle.lev_ngrams_raw(p3, t, 6)
    assert eng.comm_gather_ms() > 0
    eng.comm_set_collective(False)
    assert eng.comm_info() == (1, 0, False)
    assert _rows(eng.lev_ngrams(h, p, 2, as_array=True)) == exp
    h.release()


def test_every_search_kind_is_collective_in_a_communicator(comm_engine):
    """Round 4: no search of a context that joined a communicator returns a silent local answer.  Substitutions-only
    n-gram searches travel like the Levenshtein ones (device snapshot + ncclAllGather of records); exact searches (hit
    indices), generic searches (automaton records / folded pairs), the linear-programming routes and the has_* flags
    exchange what reached the host (two all-gathers: sizes, payload).  With one rank the merged stream is the rank's
    own — what is checked here is that every kind goes through its exchange step and comes out in the reference's
    order; the parsing / merging of 2..8 ranks' blocks is checked on the CPU (tests/test_host_logic.py)."""
    eng = comm_engine
    seq = workloads.dna(4 << 20, 78)
    pattern = workloads.dna(20, 1)
    workloads.plant_variants(seq, pattern, 256, 5)
    p, t = pattern.tobytes(), seq.tobytes()
    h = eng.upload(seq)

    def exchanged():
        ms = eng.comm_gather_ms()
        return ms

    want_subs = oracle.subs_ngrams_raw(p, t, 2)
    g0 = exchanged()
    assert _rows(eng.subs_ngrams(h, p, 2, as_array=True)) == want_subs and len(want_subs) > 50
    assert exchanged() != g0
    g0 = exchanged()
    assert eng.search_exact(h, p[:8]) == oracle.search_exact(p[:8], t)
    assert exchanged() != g0
    gen = oracle.generic_ngrams_raw(p, t, 2, 1, 1, 2)
    g0 = exchanged()
    assert eng.generic_ngrams(h, p, 2, 1, 1, 2) == gen and len(gen) > 100
    assert exchanged() != g0
    assert [r[:3] for r in eng.generic_ngrams_consolidated(h, p, 2, 1, 1, 2)] == oracle.consolidate(gen)
    assert eng.generic_ngrams_any(h, p, 2, 1, 1, 2) is True
    assert eng.subs_ngrams_any(h, p, 2) is True
    assert eng.subs_ngrams_any(h, b"T" * 20, 1) is False
    # two in flight, every kind
    eng.subs_ngrams_begin(h, p, 2)
    eng.lev_ngrams_begin(h, p, 2)
    assert eng.search_end() == want_subs and eng.search_end() == oracle.lev_ngrams_raw(p, t, 2)
    eng.generic_ngrams_begin(h, p, 2, 1, 1, 2)
    eng.generic_ngrams_begin(h, p, 2, 1, 1, 2, consolidated=True)
    assert eng.search_end() == gen
    assert [r[:3] for r in eng.search_end()] == oracle.consolidate(gen)
    h.release()
    # the linear-programming routes (short patterns) on a smaller sequence
    small = t[: 1 << 16]
    hs = eng.upload(small)
    ps = p[:5]
    assert eng.lev_lp(hs, ps, 2) == oracle.lev_lp_raw(ps, small, 2)
    assert eng.subs_lp(hs, ps, 2) == oracle.subs_lp_raw(ps, small, 2)
    assert eng.generic_lp(hs, ps, 1, 1, 1, 2) == oracle.generic_lp_raw(ps, small, 1, 1, 1, 2)
    assert eng.subs_lp_any(hs, ps, 2) is True
    hs.release()


def test_gather_capacity_follows_the_counts(comm_engine):
    """More records than the all-gather's first capacity (4 096) and than the record buffer (65 536): the search
    re-runs with a larger buffer, the gather repeats from the same snapshot with a larger capacity; afterwards a
    small search still works (capacity shrinks)."""
    eng = comm_engine
    t = (b"ACGT" * 60000)
    p = b"ACGTACGTACGT"
    h = eng.upload(t)
    exp = oracle.lev_ngrams_raw(p, t, 1)
    assert len(exp) > 70000
    assert _rows(eng.lev_ngrams(h, p, 1, as_array=True)) == exp
    eng.lev_ngrams_begin(h, p, 1)
    eng.lev_ngrams_begin(h, p, 1)
    assert _rows(eng.lev_ngrams_end(as_array=True)) == exp
    assert _rows(eng.lev_ngrams_end(as_array=True)) == exp
    h.release()
    t2 = workloads.dna(1 << 20, 3).tobytes()
    h2 = eng.upload(t2)
    for _ in range(3):
        assert _rows(eng.lev_ngrams(h2, p, 1, as_array=True)) == oracle.lev_ngrams_raw(p, t2, 1)
    h2.release()


def test_load_time_collectives_and_init_all():
    from fuzzysearch_amd import _native
    from fuzzysearch_amd import distributed as fzd
    eng = _native.Engine([0])
    try:
        with pytest.raises(ValueError):
            eng.comm_barrier()                                  # no communicator yet
        eng.comm_init_all()                                     # ncclCommInitAll over the context's devices
        assert eng.comm_info() == (1, 0, True)
        with pytest.raises(ValueError):
            eng.comm_init_all()
        shard = np.arange(100, dtype=np.uint8)
        left, right = fzd.exchange_halos_native(eng, shard, 22)
        assert len(left) == 0 and len(right) == 0
        assert eng.comm_allgather(b"hello") == [b"hello"]       # multi-device contexts hold every rank's data: single device only
        assert eng.comm_max(3.25) == 3.25
        eng.comm_barrier()
        seq = workloads.dna(1 << 20, 5)
        pattern = workloads.dna(20, 1)
        workloads.plant_variants(seq, pattern, 64, 7)
        h = eng.new_sequence(len(seq))
        eng.add_shard(h, 0, seq, 0, 0, len(seq))
        assert _rows(eng.lev_ngrams(h, pattern.tobytes(), 2, as_array=True)) == oracle.lev_ngrams_raw(pattern.tobytes(), seq.tobytes(), 2)
        h.release()
        eng.comm_destroy()
        assert eng.comm_info()[0] == 0
    finally:
        eng.close()
    dup = _native.Engine([0, 0])
    try:
        with pytest.raises(_native.UnsupportedSearch):
            dup.comm_init_all()                                 # RCCL needs one rank per GPU
    finally:
        dup.close()


def test_env_launcher_single_rank(monkeypatch):
    """init_engine_from_env with the environment a launcher sets for a one-rank job."""
    from fuzzysearch_amd import distributed as fzd
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    eng, world, rank = fzd.init_engine_from_env()
    try:
        assert (world, rank) == (1, 0) and eng.comm_info() == (1, 0, True)
    finally:
        eng.close()
