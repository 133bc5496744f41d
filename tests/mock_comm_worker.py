"""Worker of tests/test_gpu_mock_rccl.py: the N-rank collective code of libfzhip.so on ONE GPU, through the stand-in
collective library (tests/mock_rccl.cpp, named by FZ_RCCL_LIB — the caller sets it).

    python tests/mock_comm_worker.py inproc <world> [permute]        one process, <world> device states on device 0,
                                                                      fz_comm_init_all (the form bench.py --gpus N runs)
    python tests/mock_comm_worker.py rank <world> <rank> [permute]   one process per rank (the launcher form):
                                                                      fz_comm_init_rank through the rendezvous file

Every rank holds only its shard (+ halo) of every test sequence; every search is collective and must return, on every
rank, the oracle's stream of the WHOLE sequence.  `permute`: rank r owns shard perm[r] (ranks not in ownership order).
Prints "OK <checks> <records>" and exits 0, or raises."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from fuzzysearch_amd import _native  # noqa: E402
from fuzzysearch_amd import distributed as fzd  # noqa: E402
from tests import workloads  # noqa: E402

HALO = 160                     # >= m + k of every search below


def rows(arr):
    return [tuple(int(x) for x in r) for r in arr.tolist()]


class Job(object):
    """The two ways a job holds a sharded sequence: `inproc` — one engine with `world` device states; `rank` — a
    single-device engine that is one rank of `world`."""

    def __init__(self, mode, world, rank, permute):
        self.mode, self.world, self.rank = mode, world, rank
        self.perm = list(range(world))
        if permute:                                  # a fixed derangement-ish order: reversed, then rotated by one
            self.perm = self.perm[::-1]
            self.perm = self.perm[1:] + self.perm[:1]
        if mode == "inproc":
            self.eng = _native.Engine([0] * world)
            assert self.eng.comm_backend() == "stand-in", "FZ_RCCL_LIB must name tests/libmock_rccl.so"
            self.eng.comm_init_all()
            assert self.eng.comm_info() == (world, 0, True)
        else:
            self.eng = _native.Engine([0])
            assert self.eng.comm_backend() == "stand-in", "FZ_RCCL_LIB must name tests/libmock_rccl.so"
            uid = fzd.share_blob(self.eng.comm_unique_id, world, rank)
            self.eng.comm_init_rank(uid, world, rank)
            assert self.eng.comm_info() == (world, rank, True)

    def shard_args(self, t, shard, holes=()):
        n = len(t)
        lo, hi = fzd.shard_bounds(n, self.world, shard)
        if shard in holes:                           # this shard's range is given to its left neighbour instead
            return None
        while True:                                  # absorb the holes to the right
            nxt = [s for s in holes if fzd.shard_bounds(n, self.world, s)[0] == hi]
            if not nxt:
                break
            hi = fzd.shard_bounds(n, self.world, nxt[0])[1]
        b0, b1 = max(0, lo - HALO), min(n, hi + HALO)
        return np.frombuffer(t, dtype=np.uint8)[b0:b1].copy(), b0, lo, hi

    def load(self, t, holes=()):
        """-> resident handle of the sequence t (bytes), this job's shards uploaded.  `holes`: shards nobody owns as
        such (their range goes to the left neighbour), i.e. ranks that hold NOTHING of this sequence."""
        n = len(t)
        if self.mode == "inproc":
            h = self.eng.new_sequence(n)
            for r in range(self.world):
                a = self.shard_args(t, self.perm[r], holes)
                if a is not None:
                    self.eng.add_shard(h, r, a[0], a[1], a[2], a[3])
            return h
        a = self.shard_args(t, self.perm[self.rank], holes)
        if a is None:                                # an empty shard: owns [n, n)
            return self.eng.upload_shard(np.zeros(0, np.uint8), n, n, n, n)
        return self.eng.upload_shard(a[0], a[1], a[2], a[3], n)


def sharded_at_size(world, shard_mib):
    """BASELINE configs[4]'s construction (tests/test_gpu_multi_device.py: 64 copies of one block, seams, boundary plants, a
    fresh region per shard — the oracle on the block and on every window that differs) with the COLLECTIVE search: `world`
    device states on device 0 joined by fz_comm_init_all through the stand-in library, every rank's records exchanged by
    the all-gather.  The merged stream must be the complete expected multiset, block-major, every boundary plant in place."""
    from tests import test_gpu_multi_device as tmd

    def engine_cls(devices):
        eng = _native.Engine(devices)
        assert eng.comm_backend() == "stand-in"
        eng.comm_init_all()
        assert eng.comm_info() == (len(devices), 0, True)
        return eng
    n_got, n_edge = tmd._run(engine_cls, world, shard_mib, region_bytes=1 << 18)
    import ctypes
    st = (ctypes.c_uint64 * 4)()
    ctypes.CDLL(os.environ["FZ_RCCL_LIB"]).fzmock_rccl_stats(st)
    assert st[0] >= 3 and st[3] == world, list(st)
    print("OK %d %d allgathers=%d" % (n_edge, n_got, st[0]), flush=True)


def main(argv):
    if argv[0] == "sharded":
        return sharded_at_size(int(argv[1]), int(argv[2]))
    mode, world = argv[0], int(argv[1])
    rank = int(argv[2]) if mode == "rank" else 0
    permute = "permute" in argv
    job = Job(mode, world, rank, permute)
    eng = job.eng
    checks = [0, 0]

    def same(got, want, what):
        if got != want:
            gs, ws = set(got), set(want)
            raise AssertionError("%s (world %d, rank %d): %d rows, expected %d; missing %r, surplus %r, order %s" % (
                what, world, rank, len(got), len(want), sorted(ws - gs)[:4], sorted(gs - ws)[:4],
                "differs" if gs == ws else "n/a"))
        checks[0] += 1
        checks[1] += len(got)

    # ---- 1. a DNA sequence with planted variants and exact copies around every shard boundary ------------------
    n = 3 << 20
    n -= n % world
    seq = workloads.dna(n, 77)
    pattern = workloads.dna(20, 1)
    workloads.plant_variants(seq, pattern, 384, 5)
    plants = workloads.boundary_plants(20, 2, n // world, world)
    workloads.apply_plants(seq, 0, plants, pattern)
    p, t = pattern.tobytes(), seq.tobytes()
    h = job.load(t)
    exp = oracle.lev_ngrams_raw(p, t, 2)
    found = {r[:3] for r in exp}
    assert all((q, q + 20, 0) in found for q in plants)
    g0 = eng.comm_gather_ms()
    same(rows(eng.lev_ngrams(h, p, 2, as_array=True)), exp, "lev_ngrams")
    assert eng.comm_gather_ms() != g0 and eng.stats()["raw_matches"] == len(exp)
    want_subs = oracle.subs_ngrams_raw(p, t, 2)
    same(rows(eng.subs_ngrams(h, p, 2, as_array=True)), want_subs, "subs_ngrams")
    same(eng.search_exact(h, p), oracle.search_exact(p, t), "search_exact")
    same(eng.search_exact(h, p[:8]), oracle.search_exact(p[:8], t), "search_exact prefix")
    gen = oracle.generic_ngrams_raw(p, t, 2, 1, 1, 2)
    same(eng.generic_ngrams(h, p, 2, 1, 1, 2), gen, "generic_ngrams")
    same([r[:3] for r in eng.generic_ngrams_consolidated(h, p, 2, 1, 1, 2)], oracle.consolidate(gen), "generic consolidated")
    same([r[:3] for r in eng.lev_ngrams_consolidated(h, p, 2)], oracle.consolidate(exp), "lev consolidated")
    same([r[:3] for r in eng.subs_ngrams_best(h, p, 2)], [r[:3] for r in oracle.group_best(want_subs)[0]], "subs best")
    assert eng.generic_ngrams_any(h, p, 2, 1, 1, 2) is True and eng.subs_ngrams_any(h, p, 2) is True
    assert eng.subs_ngrams_any(h, b"T" * 20, 1) is False
    # wide budget (lane-per-cell verification; records with empty slots travel the same way)
    p3 = workloads.dna(48, 11).tobytes()
    same(rows(eng.lev_ngrams(h, p3, 6, as_array=True)), oracle.lev_ngrams_raw(p3, t, 6), "lev_ngrams k=6")
    # the two-deep pipeline: the gather of search i next to the scan of search i + 1, kinds mixed
    p2 = workloads.dna(24, 9).tobytes()
    exp2 = oracle.lev_ngrams_raw(p2, t, 3)
    eng.lev_ngrams_begin(h, p, 2)
    eng.lev_ngrams_begin(h, p2, 3)
    same(rows(eng.lev_ngrams_end(as_array=True)), exp, "pipeline 1")
    eng.subs_ngrams_begin(h, p, 2)
    same(rows(eng.lev_ngrams_end(as_array=True)), exp2, "pipeline 2")
    same(eng.search_end(), want_subs, "pipeline 3")
    eng.generic_ngrams_begin(h, p, 2, 1, 1, 2)       # (generic searches pipeline among themselves: two lanes)
    eng.generic_ngrams_begin(h, p, 2, 1, 1, 2, consolidated=True)
    same(eng.search_end(), gen, "pipeline 4")
    same([r[:3] for r in eng.search_end()], oracle.consolidate(gen), "pipeline 5")
    for _ in range(6):
        eng.lev_ngrams_begin(h, p, 2)
        eng.lev_ngrams_begin(h, p2, 3)
        same(rows(eng.lev_ngrams_end(as_array=True)), exp, "pipeline loop a")
        same(rows(eng.lev_ngrams_end(as_array=True)), exp2, "pipeline loop b")
    # the local form in between (the communicator stays): a one-process job still sees every shard
    eng.comm_set_collective(False)
    local = rows(eng.lev_ngrams(h, p, 2, as_array=True))
    if mode == "inproc":
        same(local, exp, "host-merged form")
    else:
        lo, hi = fzd.shard_bounds(n, world, job.perm[rank])
        assert set(local) <= set(exp) and len(local) < len(exp), "a rank's local stream is its own part only"
    eng.comm_set_collective(True)
    h.release()

    # ---- 2. capacity: ONE rank far over the all-gather's capacity (and its own record buffer), the others nearly
    # empty; then a small search again (the capacity follows the counts down) ------------------------------------
    n2 = 1 << 20
    n2 -= n2 % world
    s2 = workloads.dna(n2, 3)
    dense_shard = world // 2
    lo, hi = fzd.shard_bounds(n2, world, dense_shard)
    s2[lo:hi] = np.frombuffer(b"ACGT" * ((hi - lo) // 4 + 1), dtype=np.uint8)[:hi - lo]
    t2 = s2.tobytes()
    pd = b"ACGTACGTACGT"
    h2 = job.load(t2)
    exp_d = oracle.lev_ngrams_raw(pd, t2, 1)
    assert len(exp_d) > 70000 // max(1, world // 2)
    same(rows(eng.lev_ngrams(h2, pd, 1, as_array=True)), exp_d, "dense rank")
    eng.lev_ngrams_begin(h2, pd, 1)
    eng.lev_ngrams_begin(h2, pd, 1)
    same(rows(eng.lev_ngrams_end(as_array=True)), exp_d, "dense rank, pipeline 1")
    same(rows(eng.lev_ngrams_end(as_array=True)), exp_d, "dense rank, pipeline 2")
    same(eng.search_exact(h2, pd), oracle.search_exact(pd, t2), "dense exact")
    h2.release()
    t3 = workloads.dna(n2, 4).tobytes()
    h3 = job.load(t3)
    for _ in range(3):
        same(rows(eng.lev_ngrams(h3, pd, 1, as_array=True)), oracle.lev_ngrams_raw(pd, t3, 1), "after the dense search")
    h3.release()

    # ---- 3. ranks that hold nothing of the sequence, and a match that only a rank other than 0 holds -----------
    if world >= 3:
        holes = (1, world - 1)
        n4 = (1 << 18) - ((1 << 18) % world)
        s4 = workloads.dna(n4, 8)
        workloads.plant_variants(s4, pattern, 48, 6)
        t4 = s4.tobytes()
        h4 = job.load(t4, holes)
        same(rows(eng.lev_ngrams(h4, p, 2, as_array=True)), oracle.lev_ngrams_raw(p, t4, 2), "ranks without a shard: lev")
        same(eng.search_exact(h4, p[:6]), oracle.search_exact(p[:6], t4), "ranks without a shard: exact")
        same(eng.generic_ngrams(h4, p, 2, 1, 1, 2), oracle.generic_ngrams_raw(p, t4, 2, 1, 1, 2), "ranks without a shard: generic")
        h4.release()
    n5 = (1 << 18) - ((1 << 18) % world)
    s5 = np.full(n5, ord("A"), dtype=np.uint8)
    needle = b"CGTTGCATGCCGTAAGCTTG"
    q5 = fzd.shard_bounds(n5, world, world - 1)[0] + 1000
    s5[q5:q5 + 20] = np.frombuffer(needle, dtype=np.uint8)
    t5 = s5.tobytes()
    h5 = job.load(t5)
    assert eng.subs_ngrams_any(h5, needle, 2) is True and eng.generic_ngrams_any(h5, needle, 1, 1, 1, 2) is True
    assert eng.subs_ngrams_any(h5, b"G" * 20, 2) is False
    same(rows(eng.lev_ngrams(h5, needle, 2, as_array=True)), oracle.lev_ngrams_raw(needle, t5, 2), "needle")
    h5.release()

    # ---- 4. the linear-programming routes (short patterns), small sequence --------------------------------------
    n6 = (1 << 16) - ((1 << 16) % world)
    t6 = t[:n6]
    h6 = job.load(t6)
    ps = p[:5]
    same(eng.lev_lp(h6, ps, 2), oracle.lev_lp_raw(ps, t6, 2), "lev_lp")
    same(eng.subs_lp(h6, ps, 2), oracle.subs_lp_raw(ps, t6, 2), "subs_lp")
    same(eng.generic_lp(h6, ps, 1, 1, 1, 2), oracle.generic_lp_raw(ps, t6, 1, 1, 1, 2), "generic_lp")
    assert eng.subs_lp_any(h6, ps, 2) is True
    h6.release()

    # ---- 5. the load-time collectives (one-process-per-rank jobs) ------------------------------------------------
    if mode == "rank":
        got = eng.comm_allgather(b"rank%03d" % rank)
        assert got == [b"rank%03d" % r for r in range(world)], got
        assert eng.comm_max(float(rank) * 1.5) == (world - 1) * 1.5
        shard = np.frombuffer(t6, dtype=np.uint8)[slice(*fzd.shard_bounds(n6, world, rank))]
        left, right = fzd.exchange_halos_native(eng, shard, 22)
        lo, hi = fzd.shard_bounds(n6, world, rank)
        assert bytes(left) == t6[max(0, lo - 22):lo] and bytes(right) == t6[hi:hi + 22]
        eng.comm_barrier()
    import ctypes
    st = (ctypes.c_uint64 * 4)()
    ctypes.CDLL(os.environ["FZ_RCCL_LIB"]).fzmock_rccl_stats(st)
    assert st[0] >= 20 and st[3] == world, list(st)      # the collectives really went through the stand-in, with `world` ranks
    lib = ctypes.CDLL(os.environ["FZ_RCCL_LIB"])
    lib.fzmock_rccl_async_allgathers.restype = ctypes.c_uint64
    n_async = lib.fzmock_rccl_async_allgathers()
    if mode == "rank" and world > 1 and not os.environ.get("FZMOCK_SHM_BLOCKING"):
        # one process per rank: the all-gathers were ASYNCHRONOUS (ordered by the caller's stream only, as RCCL's are): a missing
        # dependency between the search stream and the communicator's stream could not hide behind a stream synchronise
        assert n_async >= 20, (n_async, list(st))
    eng.comm_destroy()
    eng.close()
    print("OK %d %d allgathers=%d async=%d" % (checks[0], checks[1], st[0], n_async), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
