"""-m gpu: the N-RANK collective code of libfzhip.so — gather_records, comm_gather_host, comm_rank_lows, the snapshot /
hipStreamWaitEvent ordering, the grouped all-gather, capacity regrow, the two-deep pipeline — executed with world 2, 3
and 8 on ONE GPU.  RCCL refuses two ranks on one device and no box of this pool has two GPUs, so the library's
collective table is pointed (FZ_RCCL_LIB) at a stand-in with RCCL's stream semantics (tests/mock_rccl.cpp: in-process
communicators = stream-ordered device-to-device copies, cross-process communicators = shared memory).  Everything
above the nine ncclXxx entry points is the product's real code; every rank holds only its shard (+ halo) and must
return the oracle's stream of the whole sequence.  What stays unverified is RCCL's own transport.

Every case is a subprocess: the collective library is a process-wide choice (rccl_api() is resolved once)."""
import json
import os
import subprocess
import sys
import uuid

import pytest

from tests import mock_rccl

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "mock_comm_worker.py")


def _check(proc_out, what):
    out, err, rc = proc_out
    assert rc == 0, "%s failed (rc %s)\n%s\n%s" % (what, rc, out[-2000:], err[-4000:])
    last = [ln for ln in out.splitlines() if ln.startswith("OK ")]
    assert last, out[-2000:]
    n_checks, n_rows = int(last[-1].split()[1]), int(last[-1].split()[2])
    assert n_checks >= 30 and n_rows > 10000
    return n_checks, n_rows


def _run_inproc(world, *flags):
    p = subprocess.run([sys.executable, WORKER, "inproc", str(world), *flags], capture_output=True, text=True, timeout=900,
                       env=mock_rccl.env(), cwd=ROOT)
    return _check((p.stdout, p.stderr, p.returncode), "inproc world %d %s" % (world, " ".join(flags)))


def _run_ranks(world, *flags):
    env = mock_rccl.env({"FZ_RENDEZVOUS_KEY": "mock_%s" % uuid.uuid4().hex, "WORLD_SIZE": str(world)})
    procs = []
    for r in range(world):
        e = dict(env)
        e["RANK"] = e["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, WORKER, "rank", str(world), str(r), *flags], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=e, cwd=ROOT))
    results = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=900)
            results.append((out, err, p.returncode))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [_check(res, "rank %d of %d %s" % (r, world, " ".join(flags))) for r, res in enumerate(results)]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_one_process_n_ranks(world):
    """fz_comm_init_all over `world` device states on device 0: the form the driver's `bench.py --gpus N` runs."""
    _run_inproc(world)


def test_one_process_ranks_not_in_ownership_order():
    _run_inproc(3, "permute")
    _run_inproc(8, "permute")


@pytest.mark.parametrize("world", [2, 3, 8])
def test_one_process_per_rank(world):
    """fz_comm_init_rank in `world` processes sharing device 0: the launcher form (torch.distributed.run ... bench.py)."""
    res = _run_ranks(world)
    assert len({r for r in res}) == 1                      # every rank ran the same checks and saw the same row totals


def test_one_process_per_rank_not_in_ownership_order():
    _run_ranks(3, "permute")


def test_bench_collective_line_with_eight_ranks_on_one_gpu():
    """`bench.py --gpus 8` as the driver runs it (one process, ncclCommInitAll), the eight device states on device 0 and
    the stand-in as the collective library: the line reports rccl_ranks 8 and the collective search as `value`."""
    env = mock_rccl.env({"FZ_DEVICES": ",".join(["0"] * 8)})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--mib", "128", "--steps", "12", "--warmup", "3",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["allgather_ms"] > 0
    assert d["collective_library"] == "stand-in"
    assert d["boundary_plants_found"] == 3 * 4 + 2 * 3 and d["stream_in_reference_order"] is True
    assert d["value_no_collective"] > 0 and d["value_no_collective"] != d["value"]


def _bench_launcher(world, extra_env=None, timeout=300):
    """bench.py in the launch contract's form — one process per rank, RANK / WORLD_SIZE / LOCAL_RANK in the environment —
    with `world` ranks on device 0 (LOCAL_RANK beyond the visible devices wraps: distributed.local_device).
    -> (rank 0's line, [stderr of every rank])."""
    env = mock_rccl.env(dict({"FZ_RENDEZVOUS_KEY": "mockbench_%s" % uuid.uuid4().hex, "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                              "MASTER_PORT": "29571"}, **(extra_env or {})))
    procs = []
    shm_before = set(os.listdir("/dev/shm")) if os.path.isdir("/dev/shm") else set()
    for r in range(world):
        e = dict(env)
        e["RANK"] = e["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--mib", "128", "--steps", "12",
                                       "--warmup", "3", "--settle-ms", "300", "--no-cpu-baseline"], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=e, cwd=ROOT))
    outs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=timeout)
            outs.append((out, err, p.returncode))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        if os.path.isdir("/dev/shm"):                            # a communicator the stand-in gave up on leaves its shm files behind
            for name in set(os.listdir("/dev/shm")) - shm_before:
                if name.startswith("fzmock_"):
                    try:
                        os.remove(os.path.join("/dev/shm", name))
                    except OSError:
                        pass
    for r, (out, err, rc) in enumerate(outs):
        assert rc == 0, "rank %d: %s" % (r, err[-3000:])
    lines = [ln for ln in outs[0][0].splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, outs[0][0][-2000:]
    assert not [ln for o in outs[1:] for ln in o[0].splitlines() if ln.strip().startswith("{")]      # rank 0 prints the line
    return json.loads(lines[0]), [o[1] for o in outs]


def test_launcher_form_of_bench_with_three_ranks_on_one_gpu():
    world = 3
    d, _errs = _bench_launcher(world)
    assert d["n_gpus"] == world and d["rccl_ranks"] == world and d["allgather_ms"] > 0
    assert d["stream_in_reference_order"] is True and d["boundary_plants_found"] == 3 + 2
    assert d["collective_error"] is None and d["exchange_in_timed_region"] is True


@pytest.mark.parametrize("fault,env", [
    ("the communicator cannot be set up", {"FZMOCK_FAIL_INIT": "1"}),
    ("an all-gather fails on every rank in the middle of the run", {"FZMOCK_FAIL_ALLGATHER": "30"}),
    ("an all-gather fails on ONE rank: its peers wait for it until their deadline", {"FZMOCK_FAIL_ALLGATHER": "30", "FZMOCK_FAIL_RANK": "1",
                                                                                   "FZ_COMM_TIMEOUT_MS": "400", "FZ_MOCK_RCCL_TIMEOUT_S": "3"}),
    ("rank 0 never arrives at an all-gather (deadline)", {"FZMOCK_STALL_ALLGATHER": "30:3000", "FZ_COMM_TIMEOUT_MS": "400",
                                                         "FZ_MOCK_RCCL_TIMEOUT_S": "5"}),
    ("ncclCommInitRank never returns (bench.py's watchdog; the process leaves through os._exit)", {"FZMOCK_HANG_INIT_S": "60",
                                                                                                 "FZ_COMM_INIT_TIMEOUT_S": "2"}),
])
def test_launcher_form_prints_its_line_when_the_collective_fails(fault, env):
    """The form the driver's N > 1 runs take (torch.distributed.run: one process per GPU).  Whatever the collective library
    does, the ranks agree through files to give the collective form up, search their shards without it, hand the streams
    of the last step to rank 0 through files, and rank 0 prints the line: `collective_error`, `rccl_ranks` 0, the aggregate
    `value`, all boundary plants found in the merged stream, reference order; every rank exits 0."""
    world = 3
    d, errs = _bench_launcher(world, env)
    assert d["n_gpus"] == world and d["rccl_ranks"] == 0 and d["collective_error"], (fault, d)
    assert d["value"] > 0 and d["exchange_in_timed_region"] is False
    assert d["boundary_plants_found"] == 3 + 2 and d["stream_in_reference_order"] is True     # (world 3: one odd, one even boundary)
    assert d["raw_matches"] > 0 and len(d["per_rank_ms_per_step"]) == world
    assert "per-rank form" in errs[0]
    if "HANG_INIT" in "".join(env):
        assert "joining the RCCL communicator did not return within 2 s" in d["collective_error"]
    elif "FAIL_INIT" in "".join(env):
        assert "injected failure of ncclCommInitRank" in d["collective_error"]
    elif "STALL" in "".join(env) or "FAIL_RANK" in "".join(env):
        assert "did not complete within 400 ms" in d["collective_error"]
    else:
        assert "injected failure of all-gather" in d["collective_error"]


def test_collective_search_of_eight_ranks_beyond_4gib():
    """Eight ranks x 1 GiB = 8 GiB (ranks 4 .. 7 beyond 2^32), the construction of the configs[4] test
    (tests/test_gpu_multi_device.py) with the search made collective through the stand-in: the all-gathered, merged
    stream is the complete expected multiset in block-major order with every boundary plant at its exact position."""
    p = subprocess.run([sys.executable, WORKER, "sharded", "8", "1024"], capture_output=True, text=True, timeout=1500,
                       env=mock_rccl.env(), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    last = [ln for ln in p.stdout.splitlines() if ln.startswith("OK ")]
    assert last and int(last[-1].split()[1]) == 3 * 4 + 2 * 3 and int(last[-1].split()[2]) > 1000, p.stdout[-500:]


def _bench8(extra_env, timeout=900):
    env = mock_rccl.env(dict({"FZ_DEVICES": ",".join(["0"] * 8)}, **extra_env))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--mib", "64", "--steps", "8", "--warmup", "2",
                          "--settle-ms", "200", "--no-cpu-baseline"], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]
    return json.loads(lines[0]), out.stderr


@pytest.mark.parametrize("fault,env", [
    ("init fails", {"FZMOCK_FAIL_INIT": "1"}),
    ("an all-gather returns an error in the middle of the run", {"FZMOCK_FAIL_ALLGATHER": "40"}),
    ("a rank never arrives (deadline)", {"FZMOCK_STALL_ALLGATHER": "40:4000", "FZ_COMM_TIMEOUT_MS": "300"}),
])
def test_bench_prints_its_line_when_the_collective_fails(fault, env):
    """First contact with a real 8-GPU node: whatever the collective library does — refuse the
    communicator, fail an all-gather in the middle of the timed run, never complete one — `bench.py --gpus 8` prints ITS LINE:
    `collective_error` says what happened, `value` is the host-merged search of the same shards (correct: boundary plants
    found, reference order), nothing hangs (every wait of the collective path has a deadline: FZ_COMM_TIMEOUT_MS)."""
    d, err = _bench8(env)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 0 and d["collective_error"], (fault, d.get("collective_error"))
    assert d["value"] > 0 and d["boundary_plants_found"] == 3 * 4 + 2 * 3 and d["stream_in_reference_order"] is True
    assert "host-merged" in err
    if "STALL" in "".join(env):
        assert "did not complete within 300 ms" in d["collective_error"]
    if "FAIL_ALLGATHER" in "".join(env):
        assert "injected failure" in d["collective_error"]


def test_a_late_rank_is_slow_not_wrong():
    """One rank 50 ms late in every all-gather: the collective line as usual (rccl_ranks 8, plants found), only slower."""
    d, _err = _bench8({"FZMOCK_LATE_RANK": "5:50"}, timeout=1500)
    assert d["rccl_ranks"] == 8 and d["collective_error"] is None and d["allgather_ms"] >= 40
    assert d["boundary_plants_found"] == 3 * 4 + 2 * 3 and d["stream_in_reference_order"] is True
