"""The bench line's contract (driver + judge read it): the committed profiles/r04_bench_n1.json (and r03 / r02) — the unedited
stdout of `python bench.py` on an MI355X — carries every required key with consistent values, and bench.py's
argument surface is the one the driver launches.  CPU only."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["r04_bench_n1.json", "r03_bench_n1.json", "r02_bench_n1.json"])
def test_committed_bench_line_follows_the_contract(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        text = f.read().strip()
    assert "\n" not in text, "one JSON line"
    d = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert "|p|=20" in d["metric"] and "max_l_dist=2" in d["metric"] and "|p|=20 max_l_dist=2" in base["metric"]
    assert d["unit"] == "GB/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["dtype"] == "u8" and "synthetic" in d["data"] and "workload" in d["config"]
    # value = whole-job bytes / time of exactly K steps
    shard_bytes = d["roofline"]["algorithmic_bytes_per_launch"]
    assert abs(d["value"] - shard_bytes / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 0.01
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - shard_bytes / (r["avg_kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    assert r["traffic"] is None or 0.9 < r["traffic"] * 1e9 / shard_bytes < 1.5
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "GB/s" and c["sample"]
    # the kernel cannot be slower than the step that contains it by more than the pipelining allows
    assert r["avg_kernel_ms"] <= d["ms_per_step"] * 1.02
    for block in ("target_4gib", "configs"):
        assert block in d
    if name.startswith("r04"):
        # round 4: the CPU leg's rows were compared with the GPU stream before anything was timed; the public API's time
        # stands beside the C-ABI call it wraps for all four configs
        assert c["rows_equal_gpu"] == d["raw_matches"]
        apis = [v for v in d["configs"].values() if "find_near_matches_ms" in v]
        assert len(apis) == 4 and all(v["find_near_matches_ms"] > 0 and 0.5 < v["api_over_c_abi"] < 3 for v in apis)
    if name.startswith("r03") or name.startswith("r04"):
        # round 3: the pipelined value is labelled, the one-call-at-a-time figure stands beside it
        assert "two searches in flight" in d["metric"] and d["value_sync"] < d["value"]
        assert abs(d["value_sync"] - shard_bytes / (d["sync_ms_per_call"] * 1e-3) / 1e9) / d["value_sync"] < 0.01
        assert "1024 MiB" in c["sample"] and "whole" in c["sample"]


def test_round6_line_was_measured_with_the_tree_s_native_sources():
    """profiles/r06_bench_n1.json is the unedited stdout of `python bench.py` on an MI355X with THIS tree's kernels: its
    `csrc_digest` equals the digest of fuzzysearch_amd/csrc + include/fzhip.h as they are now (the committed
    "final" line of round 5 predated the round's last product change).  And it carries the round's additions."""
    from fuzzysearch_amd import build as fzbuild
    with open(os.path.join(ROOT, "profiles", "r06_bench_n1.json")) as f:
        text = f.read().strip()
    assert "\n" not in text
    d = json.loads(text)
    assert d["csrc_digest"] == fzbuild.source_digest(), "profiles/r06_bench_n1.json predates a change of the native sources: re-run bench.py"
    r = d["roofline"]
    assert r["avg_kernel_ms"] == r["avg_kernel_ms_sync"] and r["avg_kernel_ms_pipelined"] >= r["avg_kernel_ms_sync"] * 0.98
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "200 synchronous launches" in r["avg_kernel_ms_source"]
    assert r["traffic"] is None or 0.9 < r["traffic"] * 1e9 / r["algorithmic_bytes_per_launch"] < 1.5
    assert d["settle"]["searches"] >= 64 and d["settle"]["ms"] <= 1600
    rows = {(x["m"], x["k"]): x for x in d["regimes"]["rows"]}
    assert set(rows) == {(20, 1), (20, 2), (20, 3), (20, 4), (54, 8), (100, 20)}
    assert rows[(54, 8)]["verify_form"].startswith("fused bit-vector") and rows[(54, 8)]["ms_per_call"] < 0.6
    assert rows[(20, 4)]["ms_per_call"] < 2.0 and rows[(100, 20)]["ms_per_call"] < 80.0
    assert all(abs(x["first_call_ms"] - x["ms_per_call"]) < max(0.25, 0.15 * x["ms_per_call"]) for x in rows.values()), "the form must not depend on earlier calls"
    assert "consolidated_two_in_flight_ms" in [k2 for v in d["configs"].values() for k2 in v]


def test_launcher_fallback_line_follows_the_contract():
    """The line bench.py prints under a launcher when the collective library fails (main_launcher_fallback; the committed
    sample: three ranks on one GPU through the stand-in, one rank's all-gather failing): the contract's keys, the failure said."""
    with open(os.path.join(ROOT, "profiles", "r06_bench_launcher_fallback.json")) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 3 and d["rccl_ranks"] == 0 and d["collective_error"] and d["exchange_in_timed_region"] is False
    assert d["unit"] == "GB/s" and d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    total = d["config"]["bytes_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 0.01
    assert d["ms_per_step"] == max(d["per_rank_ms_per_step"])              # the MAX over the ranks
    assert d["boundary_plants_found"] == 5 and d["stream_in_reference_order"] is True
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_two_device_states_line():
    """`FZ_DEVICES=0,0 python bench.py --gpus 2` (the torch-free N > 1 form on one GPU): n_gpus follows --gpus."""
    with open(os.path.join(ROOT, "profiles", "r03_bench_two_device_states.json")) as f:
        d = json.loads(f.read().strip())
    assert d["n_gpus"] == 2 and d["config"]["devices"] == [0, 0] and d["boundary_plants_found"] == 3
    assert d["stream_in_reference_order"] is True and len(d["kernel_ms"]["filter_per_device"]) == 2
    assert abs(d["value"] - 2 * d["config"]["bytes_per_gpu"] / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 0.01


def test_round4_multi_device_lines():
    """Round 4's N > 1 line (one torch-free process, N device states): `rccl_ranks`, `value_no_collective`, `allgather_ms`
    and the like-for-like one-GPU figure `scaling_ref_1gpu` are part of it.  Committed: two 4 GiB device states on one GPU
    (no collective: RCCL needs one rank per GPU, the line says so) and the forced collective with one rank at 4 GiB."""
    with open(os.path.join(ROOT, "profiles", "r04_bench_two_device_states.json")) as f:
        d = json.loads(f.read().strip())
    assert d["n_gpus"] == 2 and d["config"]["devices"] == [0, 0] and d["rccl_ranks"] == 0 and d["allgather_ms"] is None
    assert "not distinct" in d["config"]["sharding"] and d["value_no_collective"] == d["value"]
    assert d["boundary_plants_found"] == 3 and d["stream_in_reference_order"] is True
    r = d["scaling_ref_1gpu"]
    assert r["value"] > 0 and abs(r["x_vs_1gpu"] - d["value"] / r["value"]) < 0.01
    assert abs(d["value"] - 2 * d["config"]["bytes_per_gpu"] / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 0.01
    with open(os.path.join(ROOT, "profiles", "r04_bench_forced_collective_4gib.json")) as f:
        c = json.loads(f.read().strip())
    assert c["n_gpus"] == 1 and c["rccl_ranks"] == 1 and c["allgather_ms"] > 0 and "ncclAllGather" in c["config"]["sharding"]
    assert c["value_no_collective"] > 0 and c["scaling_ref_1gpu"]["value"] > 0
    assert abs(c["value"] - c["config"]["bytes_per_gpu"] / (c["ms_per_step"] * 1e-3) / 1e9) / c["value"] < 0.01


def test_bench_argument_surface():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
