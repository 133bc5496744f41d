"""N > 1 path on the CPU: world_size 2 and 3 over gloo (127.0.0.1 rendezvous)."""
import os
import socket
import subprocess
import sys
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture(scope="module")
def emul_so():
    out = os.path.join(tempfile.gettempdir(), "fz_hostemul_dist_%d.so" % os.getpid())
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(HERE, "host_emul.cpp"), "-o", out])
    yield out
    os.remove(out)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_allgather_equals_unsharded(world, emul_so):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "dist_worker.py"), emul_so]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = res.stdout.decode()
    assert res.returncode == 0, out[-3000:]
    assert out.count("PASS") == world, out[-3000:]


def test_shard_bounds_and_merge():
    import numpy as np
    from fuzzysearch_amd import distributed as fzd
    from tests import torch_glue
    for n in (0, 1, 7, 100, 1 << 20):
        for world in (1, 2, 3, 8):
            b = [fzd.shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
    a = [(5, 9, 0, 0), (40, 44, 1, 1)]
    c = [(105, 109, 0, 0), (101, 106, 2, 1), (130, 134, 1, 2)]
    m = fzd.merge_rank_streams([a, c, []])
    assert [tuple(r) for r in m.tolist()] == [(5, 9, 0, 0), (105, 109, 0, 0), (40, 44, 1, 1), (101, 106, 2, 1), (130, 134, 1, 2)]
    assert fzd.merge_rank_streams([[], []]).shape == (0, 4)
    # the sort-free merge of fz_match arrays gives the same order
    import random
    rnd = random.Random(3)
    for _ in range(200):
        world, nblocks = rnd.randint(1, 5), rnd.randint(1, 4)
        parts, base = [], 0
        for r in range(world):
            rows = []
            for g in range(nblocks):
                for _i in range(rnd.randint(0, 4)):
                    st = base + rnd.randint(0, 50)
                    rows.append((st, st + rnd.randint(0, 9), rnd.randint(0, 3), g))
            rows.sort(key=lambda x: x[3])
            parts.append(rows)
            base += 1000
        exp = fzd.merge_rank_streams(parts)
        got = torch_glue.merge_rank_arrays([torch_glue._as_match_array(p) for p in parts])
        assert [tuple(x) for x in got.tolist()] == [tuple(x) for x in exp.tolist()]


def _rdzv_child(args):
    rank, world, directory = args
    from fuzzysearch_amd import distributed as fzd
    first = fzd.share_blob(lambda: os.urandom(128), world, rank, timeout=60, directory=directory)
    second = fzd.share_blob(lambda: os.urandom(128), world, rank, timeout=60, directory=directory)
    return first, second


def test_torch_free_rendezvous_hands_rank0_bytes_to_every_rank():
    """share_blob: what init_engine_from_env uses to distribute the RCCL unique id (no torch, no network)."""
    import multiprocessing as mp
    world = 3
    with tempfile.TemporaryDirectory() as d:
        with mp.get_context("fork").Pool(world) as pool:
            res = pool.map(_rdzv_child, [(r, world, d) for r in range(world)], chunksize=1)
        assert len({a for a, _b in res}) == 1 and len({b for _a, b in res}) == 1
        assert res[0][0] != res[0][1] and len(res[0][0]) == 128
        assert os.listdir(d) == []                      # rank 0 cleaned up


def _file_rdzv_child(args):
    rank, world, directory = args
    import time
    from fuzzysearch_amd import distributed as fzd
    os.environ["FZ_RENDEZVOUS_KEY"] = "filerdzv_test"          # (the pool's children have different parents' views: pin the job key)
    rv = fzd.FileRendezvous(world, rank, timeout=60, directory=directory)
    time.sleep(0.01 * ((rank * 7) % 3))                         # ranks arrive in different orders
    a = rv.allgather(b"rank%d" % rank)
    rv.barrier()
    b = rv.allgather(bytes([rank]) * (1000 * (rank + 1)))       # blobs of different sizes
    for i in range(20):                                         # many quick rounds: a round's files never leak into the next
        c = rv.allgather(b"%d:%d" % (i, rank))
        assert c == [b"%d:%d" % (i, r) for r in range(world)], (i, c)
    left_behind = len([n for n in os.listdir(rv.dir) if n.endswith("_%d" % rank)])
    rv.close()
    return a, [len(x) for x in b], left_behind


def test_file_rendezvous_allgather_and_barrier():
    """FileRendezvous: what bench.py's launcher form falls back to when the collective library fails (every rank gets every
    rank's blob, in rank order; files of finished rounds are removed; close() leaves nothing behind)."""
    import multiprocessing as mp
    world = 4
    with tempfile.TemporaryDirectory() as d:
        with mp.get_context("fork").Pool(world) as pool:
            res = pool.map(_file_rdzv_child, [(r, world, d) for r in range(world)], chunksize=1)
        for a, sizes, left in res:
            assert a == [b"rank%d" % r for r in range(world)]
            assert sizes == [1000 * (r + 1) for r in range(world)]
            assert left <= 1                                    # only the last round's own file is still there before close()
        assert os.listdir(d) == []


def test_file_rendezvous_has_a_deadline():
    from fuzzysearch_amd import distributed as fzd
    with tempfile.TemporaryDirectory() as d:
        os.environ["FZ_RENDEZVOUS_KEY"] = "filerdzv_deadline"
        try:
            rv = fzd.FileRendezvous(2, 0, timeout=0.2, directory=d)
            with pytest.raises(TimeoutError, match="rank 1 never arrived"):
                rv.allgather(b"x")
        finally:
            del os.environ["FZ_RENDEZVOUS_KEY"]


def test_halos_from_edges_walks_over_short_shards():
    import numpy as np
    from fuzzysearch_amd import distributed as fzd
    seq = np.arange(40, dtype=np.uint8)
    cuts = [0, 17, 19, 20, 33, 40]                      # shards of 17, 2, 1, 13, 7 bytes; halo 5
    halo = 5
    edges = [(seq[a:b][:halo], seq[a:b][-halo:]) for a, b in zip(cuts, cuts[1:])]
    for r, (a, b) in enumerate(zip(cuts, cuts[1:])):
        left, right = fzd._halos_from_edges(edges, r, halo)
        assert left.tobytes() == seq[max(0, a - halo):a].tobytes(), r
        assert right.tobytes() == seq[b:b + halo].tobytes(), r
