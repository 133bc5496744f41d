"""-m gpu: the N > 1 forms of bench.py as far as ONE GPU allows — what the driver's `python bench.py --gpus N` runs
(one torch-free process, N device states, RCCL communicator over the context's devices), small shards.
  * FZ_BENCH_FORCE_COLLECTIVE=1, --gpus 1: ncclCommInitAll with one rank, the timed search is the collective one
    (`rccl_ranks` 1, `allgather_ms` > 0), `value_no_collective` and `scaling_ref_1gpu` from the same run;
  * FZ_DEVICES=0,0,0 --gpus 3: three device states on one GPU, per-device host worker threads, no collective
    (RCCL needs one rank per GPU: the line says so), boundary plants asserted by bench.py itself;
  * the launcher form with one rank (FZ_BENCH_FORCE_DIST=1: fz_comm_init_rank through the rendezvous code)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(env_extra, *argv):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]     # ONE JSON line (RCCL's banner goes to stderr)
    return json.loads(lines[0])


def test_forced_collective_single_device_line():
    d = _bench({"FZ_BENCH_FORCE_COLLECTIVE": "1"}, "--gpus", "1", "--mib", "256", "--steps", "20", "--warmup", "5", "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["allgather_ms"] > 0
    assert "ncclAllGather" in d["config"]["sharding"]
    assert d["value_no_collective"] > 0 and d["scaling_ref_1gpu"]["value"] > 0
    assert abs(d["value"] - d["config"]["bytes_per_gpu"] / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 0.01
    assert d["stream_in_reference_order"] is True and d["raw_matches"] > 200
    assert 0.2 < d["scaling_ref_1gpu"]["x_vs_1gpu"] < 1.5
    assert d["roofline"]["frac"] > 0.2


def test_three_device_states_on_one_gpu_line():
    d = _bench({"FZ_DEVICES": "0,0,0"}, "--gpus", "3", "--mib", "256", "--steps", "20", "--warmup", "5", "--no-cpu-baseline")
    assert d["n_gpus"] == 3 and d["rccl_ranks"] == 0 and d["allgather_ms"] is None
    assert "not distinct" in d["config"]["sharding"] and d["host_threads"] == "one worker per device"
    assert d["boundary_plants_found"] == 5 and d["stream_in_reference_order"] is True
    assert len(d["kernel_ms"]["filter_per_device"]) == 3 and min(d["kernel_ms"]["filter_per_device"]) > 0
    assert d["value_no_collective"] == d["value"]
    # the same without worker threads: the same stream (bench.py asserts plants and order), the line says which
    d2 = _bench({"FZ_DEVICES": "0,0,0", "FZ_NO_DEV_THREADS": "1"}, "--gpus", "3", "--mib", "256", "--steps", "20", "--warmup", "5", "--no-cpu-baseline")
    assert d2["host_threads"] == "calling thread only" and d2["raw_matches"] == d["raw_matches"]


def test_launcher_form_with_one_rank():
    d = _bench({"FZ_BENCH_FORCE_DIST": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "1", "--mib", "256",
               "--steps", "20", "--warmup", "5", "--no-cpu-baseline")
    assert d["rccl_ranks"] == 1 and d["allgather_ms"] > 0 and d["value_no_collective"] > 0
    assert d["scaling_ref_1gpu"]["value"] > 0


def test_collective_form_unavailable_falls_back_to_the_host_merged_form():
    """A node whose collective library cannot be loaded (FZ_NO_RCCL=1 stands for it) still gets its line: the same shards,
    per-device record lists merged on the host, `rccl_ranks` 0 and the reason in `collective_error`."""
    d = _bench({"FZ_BENCH_FORCE_COLLECTIVE": "1", "FZ_NO_RCCL": "1"}, "--gpus", "1", "--mib", "128", "--steps", "10", "--warmup", "3", "--no-cpu-baseline")
    assert d["rccl_ranks"] == 0 and d["allgather_ms"] is None and "RCCL is not available" in d["collective_error"]
    assert d["value"] > 0 and d["stream_in_reference_order"] is True and d["value_no_collective"] == d["value"]
