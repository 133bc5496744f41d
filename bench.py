#!/usr/bin/env python3
"""bench.py — GB/s of sequence scanned by the Levenshtein n-gram hot path on MI355X.

Metric (BASELINE.json): GB/s of sequence scanned (and matches/s) at |p| = 20, max_l_dist = 2.
Workload at N = 1: BASELINE configs[1] — 1 GiB of iid random DNA bytes with 1 024 planted variants
of a 20-byte pattern (tests/workloads.py::cfg2), resident in HBM before the timed
region.  One "step" = one fz_lev_ngrams() call through the C-ABI over the resident sequence: the scan
kernel (filter + fused verification, records and counters written to pinned host memory) + host
ordering of the raw match stream.

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling — every rank owns one
1 GiB shard of an N GiB global sequence, holds (m + k)-byte halos of its neighbours' bytes, scans
its shard with no data-path collective and the ranks' match lists are all-gathered over RCCL
(SURVEY.md §8(e)); the all-gather of step i runs while the scan of step i + 1 is on the GPU
(fz_lev_ngrams_begin / _end, separate HIP streams); value = N GiB / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed device settle phase before the warmup steps (a step is only ~0.35 ms, far "
                         "shorter than the GPU's DVFS ramp: 23 back-to-back steps run the kernel 12%% slower "
                         "than 250; the timed region is unaffected: exactly --steps steps)")
    ap.add_argument("--mib", type=int, default=1024, help="MiB of sequence per GPU (default: 1 GiB = configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=1024)
    return ap.parse_args()


def cpu_baseline(seq, pattern, k, sample_mib):
    """Time the reference's own native path (oracle/_ref) — or the C port — on a bounded sample."""
    import oracle
    n = min(len(seq), sample_mib << 20)
    sample = seq[:n].tobytes()
    p = pattern.tobytes()
    kind = "port"
    try:
        from oracle import ref_glue, ref_loader
        if ref_loader.have_ref_natives():
            fn = lambda: ref_glue.lev_ngrams_raw(p, sample, k)     # noqa: E731
            fn_small = lambda: ref_glue.lev_ngrams_raw(p, sample[:1 << 20], k)   # noqa: E731
            fn_small()
            kind = "reference"
    except Exception as exc:                                        # pragma: no cover
        print("cpu_baseline: reference natives unusable (%r); timing the C port instead" % (exc,), file=sys.stderr)
        kind = "port"
    if kind == "port":
        fn = lambda: oracle.lev_ngrams_raw(p, sample, k)           # noqa: E731
    t0 = time.perf_counter()
    res = fn()
    dt = time.perf_counter() - t0
    return {"value": n / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": kind,
            "sample": "first %d MiB of the same DNA workload, |p|=20 k=2, 1 pass, single thread "
                      "(the reference has no parallelism); %d raw matches in %.2f s" % (n >> 20, len(res), dt)}


def measured_traffic(shard_bytes):
    """HBM bytes per launch of the scan kernel from the committed PMC profile (FETCH_SIZE with the
    gfx950 x2 correction + WRITE_SIZE, separate rocprofv3 --pmc passes: benchmarks/profile.sh ->
    profiles/*_pmc_summary.json).  Only reported for the workload size the profile was taken at."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get("algorithmic_bytes_per_launch") == shard_bytes:
                return round(d["hbm_traffic"]["traffic_bytes"] / 1e9, 4), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    # FZ_BENCH_FORCE_DIST=1: take the RCCL code path with a single rank too (1-GPU smoke of it)
    use_dist = world > 1 or os.environ.get("FZ_BENCH_FORCE_DIST") == "1"
    if use_dist:
        # import torch BEFORE libfzhip so both share one HIP runtime (same libamdhip64 SONAME)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from fuzzysearch_amd import _native
    from fuzzysearch_amd import distributed as fzd
    from tests import workloads

    k = 2
    shard_bytes = args.mib << 20
    pattern = workloads.dna(20, 1)
    m = len(pattern)
    halo = m + k
    p = pattern.tobytes()
    global_n = shard_bytes * world

    # ---- synthetic data: shard r of the global sequence ---------------------------------------
    if world == 1:
        seq, _, _ = workloads.cfg2(shard_bytes, 1024)
    else:
        seq = workloads.dna(shard_bytes, 20250925 + rank)
        workloads.plant_variants(seq, pattern, 1024, 7 + rank)
        # variants straddling the boundary with the next shard: first half lives in our tail
        seq[-10:] = pattern[:10]
        if rank > 0:
            seq[:10] = pattern[10:]

    engine = _native.Engine([local_rank])
    if not use_dist:
        handle = engine.upload(seq)
    else:
        # halo exchange, once at load: neighbours' (m + k) edge bytes (RCCL all_gather)
        left, right = fzd.exchange_halos(seq, halo)
        buf = np.concatenate([left, seq, right])
        own_lo = rank * shard_bytes
        handle = engine.upload_shard(buf, own_lo - len(left), own_lo, own_lo + shard_bytes, global_n)
        del buf

    def sync():
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        # synchronous: on return the ordered raw match stream is on the host (numpy view of the
        # C-ABI's fz_match array; no per-record Python objects inside the timed region)
        raw = engine.lev_ngrams(handle, p, k, as_array=True)
        if use_dist:
            return fzd.allgather_matches(raw, as_array=True)   # ONE RCCL all_gather: counts + packed records
        return raw

    # setup self-check + clock settle (untimed): repeated searches must return the identical stream
    t_settle = time.perf_counter()
    first = step()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        again = step()
        assert len(again) == len(first), "non-deterministic result"
    for _ in range(args.warmup):
        matches = step()
    filter_ms, verify_ms, device_ms = [], [], []
    sync()
    t0 = time.perf_counter()
    if not use_dist:
        for _ in range(args.steps):
            matches = step()
            f_, v_, d_ = engine.kernel_ms()                # hipEvent spans of this step's kernels
            filter_ms.append(f_)
            verify_ms.append(v_)
            device_ms.append(d_)
    else:
        # N > 1: the collective of step i overlaps the scan of step i + 1 (fz_lev_ngrams_begin / _end;
        # the scan runs on the engine's own HIP stream, RCCL and its staging copies on torch's).  Every
        # step still ends with its merged, ordered match list on this rank's host, and all K searches
        # and K all-gathers complete inside the timed region.
        engine.lev_ngrams_begin(handle, p, k)
        for i in range(args.steps):
            raw = engine.lev_ngrams_end(as_array=True)
            f_, v_, d_ = engine.kernel_ms()
            filter_ms.append(f_)
            verify_ms.append(v_)
            device_ms.append(d_)
            if i + 1 < args.steps:
                engine.lev_ngrams_begin(handle, p, k)
            matches = fzd.allgather_matches(raw, as_array=True)
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    st = engine.stats()
    if rank == 0:
        import fuzzysearch_amd as fa
        matches = [tuple(int(x) for x in r) for r in matches.tolist()]
        consolidated = fa.common._native.consolidate(matches)
        ms_per_step = elapsed / args.steps * 1e3
        value = global_n * args.steps / elapsed / 1e9
        f_ms = float(np.mean(filter_ms))
        achieved = shard_bytes / (f_ms * 1e-3) / 1e9           # algorithmic bytes: N read once
        traffic_gb, traffic_src = measured_traffic(shard_bytes)
        out = {
            "metric": "GB/s of sequence scanned at |p|=20 max_l_dist=2 (levenshtein_ngram path)",
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%d MiB iid random DNA bytes per GPU, |pattern|=20, max_l_dist=2, "
                                   "1024 planted variants per GiB (BASELINE configs[1]); resident in HBM" % args.mib,
                       "bytes_per_gpu": shard_bytes, "pattern_len": m, "max_l_dist": k,
                       "sharding": "none" if not use_dist else "contiguous shards, (m+k)-byte halo, RCCL all_gather of "
                                   "matches; the all_gather of step i overlaps the scan of step i+1"},
            "matches_per_s": round(len(matches) * args.steps / elapsed, 1),
            "raw_matches": len(matches),
            "consolidated_matches": len(consolidated),
            "ngram_hits": st["ngram_hits"],
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic_gb,
                         "traffic_unit": "GB per launch (rocprofv3 PMC, committed profile)", "traffic_source": traffic_src,
                         "kernel": "fz_scan_kernel", "avg_kernel_ms": round(f_ms, 4),
                         "algorithmic_bytes_per_launch": shard_bytes},
            "kernel_ms": {"filter": round(f_ms, 4), "verify": round(float(np.mean(verify_ms)), 4),
                          "device_total": round(float(np.mean(device_ms)), 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(seq, pattern, k, args.cpu_sample_mib)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
