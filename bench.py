#!/usr/bin/env python3
"""bench.py — GB/s of sequence scanned by the Levenshtein n-gram hot path on MI355X.

Metric (BASELINE.json): GB/s of sequence scanned (and matches/s) at |p| = 20, max_l_dist = 2.
Workload at N = 1: BASELINE configs[1] — 1 GiB of iid random DNA bytes with 1 024 planted variants
of a 20-byte pattern (tests/workloads.py::cfg2), resident in HBM before the timed
region.  One "step" = one fz_lev_ngrams() call through the C-ABI over the resident sequence: the scan
kernel (filter + fused verification, records and counters written to pinned host memory) + host
ordering of the raw match stream.

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling, BASELINE configs[4] — every
rank owns one 4 GiB shard of a 4N GiB global sequence (32 GiB at N = 8; copies of the pattern planted
around every shard boundary), holds (m + k)-byte halos of its neighbours' bytes, scans its shard with
no data-path collective and the ranks' match lists are all-gathered over RCCL; the
all-gather of step i runs while the scan of step i + 1 is on the GPU (fz_lev_ngrams_begin / _end, two
searches in flight, separate HIP streams); value = 4N GiB / max-over-ranks time.

`python bench.py --gpus N` WITHOUT a launcher (what the driver runs for N > 1 on one node): one torch-free process, N device
states joined into an RCCL communicator (ncclCommInitAll), the collective search timed as `value` (main_multi_device); if the
communicator cannot be set up the same shards are searched without it and the line says so (`collective_error`).

Next to the contract's fields the N = 1 line carries `roofline`, `cpu_baseline` (the reference's natives, rows compared
with the GPU stream before timing), `target_4gib` (the north-star size) and `configs` — the other BASELINE configs at 1 GiB
and, for configs[1], `end_to_end` (the reference's call form on plain bytes: first / repeat / cache-off call, PCIe
included, never `value`) and `file_api` (find_near_matches_in_file on the same GiB).  A failure in those secondary blocks
is reported in the line (`extras_error`, `end_to_end_error`, `file_api.error`), never instead of it.

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
    ap.add_argument("--settle-ms", type=float, default=1500.0,
                    help="cap of the untimed, adaptive device settle phase before the warmup steps (a step is only ~0.3 ms, far "
                         "shorter than the GPU's DVFS ramp; it ends when the kernel's span has stopped moving; the timed region "
                         "is unaffected: exactly --steps steps)")
    ap.add_argument("--settle-min-ms", type=float, default=100.0, help="shortest settle phase")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure the scan kernel's HBM traffic in this run (two rocprofv3 --pmc child passes, ~1 min); "
                         "roofline.traffic then comes from the committed profile")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--mib", type=int, default=0, help="MiB of sequence per GPU (default: 1024 = BASELINE configs[1] at N = 1, "
                    "4096 = configs[4] — 32 GiB over 8 GPUs — at N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=1024)
    ap.add_argument("--no-extras", action="store_true", help="skip the 4 GiB target and the other BASELINE configs (N = 1 extras)")
    ap.add_argument("--sync", action="store_true", help="time synchronous fz_lev_ngrams calls (one search in flight) instead of the two-deep pipeline")
    ap.add_argument("--two-streams", action="store_true",
                    help="after the timed region, also time the two-deep pipeline with the younger scan on a second stream "
                         "(fz_set_streams(2)) and report it as `two_streams` beside `value`.  Not part of the default run: the "
                         "scans then overlap, and a kernel trace of the run would average overlapped launches of the headline kernel")
    return ap.parse_args()


def _cpu_fn():
    """-> (callable(p, t, k) -> raw matches, kind): the reference's own natives (oracle/_ref) or the C port."""
    import oracle
    try:
        from oracle import ref_glue, ref_loader
        if ref_loader.have_ref_natives():
            ref_glue.lev_ngrams_raw(b"ACGTACGTACGT", b"ACGTACGTACGTACGT" * 64, 1)
            return ref_glue.lev_ngrams_raw, "reference"
    except Exception as exc:                                        # pragma: no cover
        print("cpu_baseline: reference natives unusable (%r); timing the C port instead" % (exc,), file=sys.stderr)
    return oracle.lev_ngrams_raw, "port"


_CPU_SHARD = {}


def _cpu_worker(args):
    lo, hi = args
    fn, t, p, k = _CPU_SHARD["fn"], _CPU_SHARD["t"], _CPU_SHARD["p"], _CPU_SHARD["k"]
    return [(s + lo, e + lo, d) for (s, e, d, *_r) in fn(p, t[lo:hi], k)]


def cpu_baseline(seq, pattern, k, sample_mib, one_core_mib=1024):
    """The reference CPU path on the host cores of this box: (i) one core, as shipped
    (the reference has no parallelism), best of 3 on a bounded sample; (ii) all cores: the sample cut into
    os.cpu_count() contiguous shards overlapping by m - 1 + k bytes (the reference's own chunk overlap,
    __init__.py:135-138), one process per shard (fork, before the HIP runtime exists), best of 3.
    Must run BEFORE the GPU engine is created (fork)."""
    import multiprocessing as mp
    fn, kind = _cpu_fn()
    p = pattern.tobytes()
    n1 = min(len(seq), one_core_mib << 20)
    t1 = seq[:n1].tobytes()
    best1, res1 = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        res1 = fn(p, t1, k)
        dt = time.perf_counter() - t0
        best1 = dt if best1 is None else min(best1, dt)
    out = {"value": round(n1 / best1 / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": kind,
           "sample": "%s %d MiB of the same DNA workload, |p|=20 k=2, single thread (the reference has no "
                     "parallelism), best of 3; %d raw matches in %.2f s" % ("the whole" if n1 == len(seq) else "first", n1 >> 20, len(res1), best1)}
    cores = os.cpu_count() or 1
    n = min(len(seq), sample_mib << 20)
    t = seq[:n].tobytes()
    keep = len(p) - 1 + k
    bounds = [(max(0, n * i // cores - keep), min(n, n * (i + 1) // cores)) for i in range(cores)]
    _CPU_SHARD.update(fn=fn, t=t, p=p, k=k)
    try:
        ctx = mp.get_context("fork")
        with ctx.Pool(cores) as pool:
            pool.map(_cpu_worker, [(0, min(n, 1 << 16))] * cores)          # start the workers (untimed)
            best, merged = None, None
            for _ in range(3):
                t0 = time.perf_counter()
                parts = pool.map(_cpu_worker, bounds, chunksize=1)
                merged = [m for part in parts for m in part]
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
        out["all_cores"] = {"value": round(n / best / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": kind,
                            "sample": "first %d MiB, %d processes over contiguous shards with %d-byte overlap, matches "
                                      "merged; best of 3 (%.2f s, %d raw matches incl. overlap duplicates)"
                                      % (n >> 20, cores, keep, best, len(merged))}
    except Exception as exc:                                        # pragma: no cover
        out["all_cores"] = {"error": repr(exc), "cores": cores}
    finally:
        _CPU_SHARD.clear()
    out["_rows"] = [tuple(r[:3]) for r in res1]          # (popped by the caller: compared with the GPU stream, not printed)
    out["_rows_bytes"] = n1
    return out


def time_call(engine, fn, reps, warm_s=0.2):
    """-> (ms per call at the C-ABI, mean scan kernel ms, mean verify kernel ms, last result)."""
    t_end = time.perf_counter() + warm_s
    res = fn()
    while time.perf_counter() < t_end:
        res = fn()
    f_ms, v_ms = [], []
    t0 = time.perf_counter()
    for _ in range(reps):
        res = fn()
        f_, v_, _d = engine.kernel_ms()
        f_ms.append(f_)
        v_ms.append(v_)
    dt = (time.perf_counter() - t0) / reps
    return dt * 1e3, float(np.mean(f_ms)), float(np.mean(v_ms)), res


def time_pipelined(engine, handle, p, k, reps, want):
    """ms per search with two searches in flight (fz_lev_ngrams_begin / _end), as the headline step runs; the last
    two collected streams are compared with `want` (the synchronous call's result)."""
    engine.lev_ngrams_begin(handle, p, k)
    for _ in range(10):                                         # warm-up, keeps one search in flight
        engine.lev_ngrams_begin(handle, p, k)
        engine.lev_ngrams_end(as_array=True)
    t0 = time.perf_counter()
    for _ in range(reps):                                       # one search starts and one completes per iteration
        engine.lev_ngrams_begin(handle, p, k)
        raw = engine.lev_ngrams_end(as_array=True)
    dt = (time.perf_counter() - t0) / reps
    raw_last = engine.lev_ngrams_end(as_array=True)
    assert np.array_equal(raw, want) and np.array_equal(raw_last, want), "pipelined search returned a different stream"
    return dt * 1e3


def time_api(fn, reps, warm_s=0.2):
    """Mean ms of a public-API call (find_near_matches on a resident sequence: search + consolidation + Match objects)."""
    t_end = time.perf_counter() + warm_s
    res = fn()
    while time.perf_counter() < t_end:
        res = fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = fn()
    return (time.perf_counter() - t0) / reps * 1e3, res


def api_block(fa, engine, seq, kwargs, pattern, c_abi_ms, reps):
    """find_near_matches(pattern, resident(bytes), **kwargs) next to the C-ABI call it wraps: the sequence handed over as
    `bytes`, as the reference is called (Match.matched slices are bytes objects)."""
    data = seq.tobytes()
    res = fa.resident(data, engine=engine)
    ms, out = time_api(lambda: fa.find_near_matches(pattern, res, **kwargs), reps)
    res.release()
    return {"find_near_matches_ms": round(ms, 4), "api_over_c_abi": round(ms / c_abi_ms, 3), "api_matches": len(out)}


def extra_blocks(engine, workloads, reps):
    """Driver-visible numbers for the north-star target (4 GiB DNA) and the other BASELINE configs, measured
    in the same run as the headline (N = 1 only): C-ABI GB/s, kernel ms, raw match counts."""
    import fuzzysearch_amd as fa
    out = {}
    pattern = workloads.dna(20, 1)
    p = pattern.tobytes()
    gib = 1 << 30
    # north star: 4 GiB DNA, |p| = 20, k = 2
    seq = np.empty(4 * gib, dtype=np.uint8)
    for i in range(4):
        seq[i * gib:(i + 1) * gib] = workloads.dna(gib, 20250925 + i)
    workloads.plant_variants(seq, pattern, 4096, 7)
    h = engine.upload(seq)
    ms, f_ms, v_ms, res = time_call(engine, lambda: engine.lev_ngrams(h, p, 2, as_array=True), reps)
    st = engine.stats()
    pipe_ms = time_pipelined(engine, h, p, 2, reps, res)
    h.release()
    del seq
    out["target_4gib"] = {"workload": "4 GiB iid random DNA bytes, |pattern|=20, max_l_dist=2, 4096 planted variants; resident",
                          "ms_per_call": round(ms, 4), "GB_per_s": round(4 * gib / ms / 1e6, 1),
                          "two_in_flight_ms_per_call": round(pipe_ms, 4), "two_in_flight_GB_per_s": round(4 * gib / pipe_ms / 1e6, 1),
                          "scan_kernel_ms": round(f_ms, 4),
                          "roofline_frac": round(4 * gib / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "raw_matches": int(len(res)), "ngram_hits": int(st["ngram_hits"])}
    cfgs = {}
    seq, pat, _ = workloads.cfg3(gib, 1024)
    p2 = pat.tobytes()
    h = engine.upload(seq)
    ms, f_ms, v_ms, res = time_call(engine, lambda: engine.subs_ngrams(h, p2, 3, as_array=True), reps)
    h.release()
    cfgs["configs[2] ASCII m=32 subs<=3 (substitutions_only)"] = {
        "ms_per_call": round(ms, 4), "GB_per_s": round(gib / ms / 1e6, 1), "scan_kernel_ms": round(f_ms, 4), "raw_matches": int(len(res))}
    cfgs["configs[2] ASCII m=32 subs<=3 (substitutions_only)"].update(
        api_block(fa, engine, seq, dict(max_substitutions=3, max_insertions=0, max_deletions=0), p2, ms, reps))
    seq, pat, _ = workloads.cfg4(gib, 1024)
    p3 = pat.tobytes()
    h = engine.upload(seq)
    ms, f_ms, v_ms, res = time_call(engine, lambda: engine.lev_ngrams(h, p3, 5, as_array=True), reps)
    pipe_ms = time_pipelined(engine, h, p3, 5, reps, res)
    cfgs["configs[3a] UTF-8 m=64 max_l_dist=5 (levenshtein_ngram, lane-per-cell verify fused into the scan)"] = {
        "ms_per_call": round(ms, 4), "GB_per_s": round(gib / ms / 1e6, 1),
        "two_in_flight_ms_per_call": round(pipe_ms, 4), "two_in_flight_GB_per_s": round(gib / pipe_ms / 1e6, 1),
        "scan_kernel_ms": round(f_ms, 4),
        "verify_kernel_ms": round(v_ms, 4), "raw_matches": int(len(res))}
    cfgs["configs[3a] UTF-8 m=64 max_l_dist=5 (levenshtein_ngram, lane-per-cell verify fused into the scan)"].update(
        api_block(fa, engine, seq, dict(max_l_dist=5), p3, ms, reps))
    ms, f_ms, v_ms, res = time_call(engine, lambda: engine.generic_ngrams(h, p3, 5, 2, 2, 5, as_array=True), max(20, reps // 4))
    cms, _f, cv_ms, cres = time_call(engine, lambda: engine.generic_ngrams_consolidated(h, p3, 5, 2, 2, 5, as_array=True), max(20, reps // 4))
    # two generic searches in flight (two lanes: the scan of one next to the automaton kernel of the other)
    g_reps = max(20, reps // 4)
    engine.generic_ngrams_begin(h, p3, 5, 2, 2, 5)
    for _ in range(5):
        engine.generic_ngrams_begin(h, p3, 5, 2, 2, 5)
        engine.search_end(as_array=True)
    t0 = time.perf_counter()
    for _ in range(g_reps):
        engine.generic_ngrams_begin(h, p3, 5, 2, 2, 5)
        graw = engine.search_end(as_array=True)
    gpipe_ms = (time.perf_counter() - t0) / g_reps * 1e3
    glast = engine.search_end(as_array=True)
    assert np.array_equal(graw, res) and np.array_equal(glast, res), "pipelined generic search returned a different stream"
    # ... and two CONSOLIDATED generic searches in flight (what find_near_matches runs for a GenericSearch)
    engine.generic_ngrams_begin(h, p3, 5, 2, 2, 5, consolidated=True)
    for _ in range(5):
        engine.generic_ngrams_begin(h, p3, 5, 2, 2, 5, consolidated=True)
        engine.search_end(as_array=True)
    t0 = time.perf_counter()
    for _ in range(g_reps):
        engine.generic_ngrams_begin(h, p3, 5, 2, 2, 5, consolidated=True)
        craw = engine.search_end(as_array=True)
    cpipe_ms = (time.perf_counter() - t0) / g_reps * 1e3
    clast = engine.search_end(as_array=True)
    assert np.array_equal(craw, cres) and np.array_equal(clast, cres), "pipelined consolidated generic search returned different rows"
    h.release()
    api3b = api_block(fa, engine, seq, dict(max_substitutions=5, max_insertions=2, max_deletions=2, max_l_dist=5), p3, ms, max(20, reps // 4))
    cfgs["configs[3b] UTF-8 m=64 limits (5,2,2,5) (generic_search)"] = {
        "ms_per_call": round(ms, 4), "GB_per_s": round(gib / ms / 1e6, 1), "scan_kernel_ms": round(f_ms, 4),
        "automaton_kernel_ms": round(v_ms, 4), "raw_matches": int(len(res)),
        "two_in_flight_ms_per_call": round(gpipe_ms, 4), "two_in_flight_GB_per_s": round(gib / gpipe_ms / 1e6, 1),
        "consolidated_ms_per_call": round(cms, 4), "consolidated_GB_per_s": round(gib / cms / 1e6, 1),
        "consolidated_automaton_kernel_ms": round(cv_ms, 4), "consolidated_matches": int(len(cres)),
        "consolidated_two_in_flight_ms": round(cpipe_ms, 4), "consolidated_two_in_flight_GB_per_s": round(gib / cpipe_ms / 1e6, 1),
        "note": "consolidated = fz_generic_ngrams_consolidated: search + consolidate_overlapping_matches, first stage on the device"}
    cfgs["configs[3b] UTF-8 m=64 limits (5,2,2,5) (generic_search)"].update(api3b)
    # configs[1] through the public API (the headline workload): find_near_matches on a resident bytes sequence
    seq, pat, _ = workloads.cfg2(gib, 1024)
    p1 = pat.tobytes()
    h = engine.upload(seq)
    ms, f_ms, v_ms, res = time_call(engine, lambda: engine.lev_ngrams(h, p1, 2, as_array=True), reps)
    h.release()
    cfgs["configs[1] DNA m=20 max_l_dist=2 (levenshtein_ngram) through find_near_matches"] = {
        "ms_per_call": round(ms, 4), "GB_per_s": round(gib / ms / 1e6, 1), "scan_kernel_ms": round(f_ms, 4), "raw_matches": int(len(res))}
    cfgs["configs[1] DNA m=20 max_l_dist=2 (levenshtein_ngram) through find_near_matches"].update(
        api_block(fa, engine, seq, dict(max_l_dist=2), p1, ms, reps))
    try:
        out["regimes"] = regimes_block(engine, workloads, seq)
    except Exception as exc:  # noqa: BLE001
        out["regimes"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    try:
        cfgs["configs[1] DNA m=20 max_l_dist=2 (levenshtein_ngram) through find_near_matches"].update(
            end_to_end_block(fa, seq, p1, reps))
    except Exception as exc:  # noqa: BLE001 — the secondary numbers must not cost the run its headline line
        cfgs["configs[1] DNA m=20 max_l_dist=2 (levenshtein_ngram) through find_near_matches"]["end_to_end_error"] = "%s: %s" % (type(exc).__name__, exc)
    out["configs"] = cfgs
    return out


FORM_NAMES = ["none", "fused register band", "fused lane-per-cell", "fused bit-vector (64-bit columns)",
              "fused bit-vector (128-bit columns)", "stand-alone kernel", "fused bit-vector (32-bit columns)"]


def _csrc_digest():
    """Which native sources this line was measured with (fuzzysearch_amd/build.py: source_digest)."""
    try:
        from fuzzysearch_amd import build as fzbuild
        return fzbuild.source_digest()
    except Exception:  # noqa: BLE001
        return None


def regimes_block(engine, workloads, seq):
    """The verification cliff map: the Levenshtein n-gram search on the SAME GiB of DNA as configs[1]
    with |p| = 20, k = 1 .. 4 (3.3e3 .. 2.1e7 candidates), |p| = 54, k = 8 (2.4e6) and |p| = 100, k = 20 (1.0e8: four-
    character n-grams, one hit per 10 bytes) — ms per synchronous C-ABI call, GB/s, verified candidates, and the verification
    form, which is a function of the search's arguments alone (first call == steady state: `first_ms`)."""
    rows = []
    h = engine.upload(seq)
    gib = len(seq)
    for m, k in [(20, 1), (20, 2), (20, 3), (20, 4), (54, 8), (100, 20)]:
        p = workloads.dna(m, 7 if m != 20 else 1).tobytes()
        t0 = time.perf_counter()
        first = engine.lev_ngrams(h, p, k, as_array=True)
        first_ms = (time.perf_counter() - t0) * 1e3
        slow = first_ms > 20.0
        ms, f_ms, v_ms, res = time_call(engine, lambda: engine.lev_ngrams(h, p, k, as_array=True), 3 if slow else 20,
                                        warm_s=0.0 if slow else 0.1)
        st = engine.stats()
        assert np.array_equal(first, res), "the first and the later calls returned different streams"
        rows.append({"m": m, "k": k, "ms_per_call": round(ms, 4), "GB_per_s": round(gib / ms / 1e6, 1), "first_call_ms": round(first_ms, 3),
                     "kernel_ms": round(f_ms + v_ms, 4), "candidates": int(st["ngram_hits"]), "raw_matches": int(len(res)),
                     "verify_form": FORM_NAMES[int(st["verify_form"])], "scan_launches": int(st["filter_launches"])})
    h.release()
    return {"workload": "configs[1]'s 1 GiB of DNA; Levenshtein n-gram search, resident, one synchronous C-ABI call", "rows": rows}


def end_to_end_block(fa, seq, p1, reps):
    """"also report end-to-end (incl. H2D and Python Match construction) separately" — the calls the
    reference's users make, on configs[1] (1 GiB DNA, |p| = 20, k = 2), NEVER `value`:
      end_to_end: find_near_matches(p, <bytes>) exactly as the reference is called (__init__.py:35-57).  `first_call` = a
        bytes object the library has not seen: pageable H2D upload + search + consolidation + Match objects; `repeat_call` =
        the same object again: the residency cache (engine.ResidencyCache) finds it in HBM, nothing crosses PCIe;
        `cache_off_call` = the round-4 behaviour (every call uploads).
      file_api: find_near_matches_in_file (__init__.py:86-171) on the same GiB as a file in the page cache (/dev/shm), the
        reference's default 1 MiB chunks, through the streaming pipeline; result compared with the in-memory search where
        the chunk geometry allows (matches that do not straddle a chunk seam are identical)."""
    import tempfile
    from fuzzysearch_amd import engine as fzengine
    cache = fzengine.residency_cache()
    cache.clear()
    nbytes = len(seq)
    data = seq.tobytes()
    kwargs = dict(max_l_dist=2)
    t0 = time.perf_counter()
    first = fa.find_near_matches(p1, data, **kwargs)
    first_ms = (time.perf_counter() - t0) * 1e3
    info1 = cache.info()
    rep_ms, rep = time_api(lambda: fa.find_near_matches(p1, data, **kwargs), reps)
    info2 = cache.info()
    assert rep == first, "repeat call on a resident sequence returned different matches"
    assert info2["misses"] == info1["misses"], "the repeat calls uploaded the sequence again"
    # a second, never-seen bytes object of the same content: first-call cost once more (median of three fresh objects)
    fresh = []
    for _ in range(3):
        cache.clear()
        d2 = bytes(memoryview(data))                               # a new object: the cache does not know it
        t0 = time.perf_counter()
        r2 = fa.find_near_matches(p1, d2, **kwargs)
        fresh.append((time.perf_counter() - t0) * 1e3)
        assert r2 == first
        del d2
    first_ms = sorted(fresh + [first_ms])[1]
    old_budget = cache.budget
    cache.clear()
    cache.budget = 0
    try:
        off_ms, off = time_api(lambda: fa.find_near_matches(p1, data, **kwargs), 3, warm_s=0.0)
    finally:
        cache.budget = old_budget
    assert off == first
    out = {"end_to_end": {
        "call": "find_near_matches(pattern, <bytes>, max_l_dist=2) — the reference's call form, host buffer handed over every call",
        "first_call_ms": round(first_ms, 2), "first_call_GB_per_s": round(nbytes / first_ms / 1e6, 1),
        "repeat_call_ms": round(rep_ms, 4), "repeat_call_GB_per_s": round(nbytes / rep_ms / 1e6, 1),
        "cache_off_call_ms": round(off_ms, 2), "cache_off_GB_per_s": round(nbytes / off_ms / 1e6, 1),
        "matches": len(first),
        "note": "first_call includes the pageable H2D upload (PCIe) and is never `value`; repeat_call: the same immutable object is "
                "found resident (FUZZYSEARCH_HIP_RESIDENT_CACHE, default 8G; bytes / str only)"}}
    cache.clear()
    d = "/dev/shm" if os.path.isdir("/dev/shm") else None
    if d is not None:
        try:                                   # (a tmpfs too small for the GiB: the default temporary directory)
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize < nbytes + (64 << 20):
                d = None
        except OSError:
            d = None
    name = None
    try:
        with tempfile.NamedTemporaryFile(delete=False, dir=d) as f:
            f.write(data)
            name = f.name
        best, res = None, None
        for _ in range(3):
            with open(name, "rb") as f:
                t0 = time.perf_counter()
                res = fa.find_near_matches_in_file(p1, f, **kwargs)
                dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        chunk, keep = 1 << 20, len(p1) - 1 + 2
        stride = chunk - keep
        # away from the chunk seams the file search and the in-memory search see the same bytes around a match
        def inner(ms):
            return [(x.start, x.end, x.dist) for x in ms if (x.start % stride) > 2 * len(p1) and (x.end % stride) < stride - 2 * len(p1)]
        same = inner(res) == inner(first)
        out["file_api"] = {
            "call": "find_near_matches_in_file(pattern, open(<1 GiB file in the page cache>, 'rb'), max_l_dist=2), _chunk_size 2**20 (the reference's default)",
            "seconds": round(best, 4), "GB_per_s": round(nbytes / best / 1e9, 2), "matches": len(res),
            "matches_equal_in_memory_away_from_chunk_seams": bool(same), "in_memory_matches": len(first),
            "note": "best of three; page cache -> pinned staging (pread pool) -> H2D -> scan with per-chunk clamps; chunk geometry of the reference kept"}
    except Exception as exc:  # noqa: BLE001
        out["file_api"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    finally:
        if name:
            try:
                os.remove(name)
            except OSError:
                pass
    return out


def traffic_child(args):
    """`bench.py --traffic-child --mib M` (run under `rocprofv3 --pmc ...` by live_traffic): the headline workload, a few
    searches, nothing printed."""
    from fuzzysearch_amd import _native
    from tests import workloads
    seq, pattern, _ = workloads.cfg2((args.mib or 1024) << 20, 1024)
    engine = _native.Engine([0])
    h = engine.upload(seq)
    for _ in range(12):
        engine.lev_ngrams(h, pattern.tobytes(), 2, as_array=True)
    h.release()


def live_traffic(mib):
    """HBM bytes per launch of the scan kernel, measured NOW: two child runs of this script under `rocprofv3 --pmc FETCH_SIZE`
    and `--pmc WRITE_SIZE` (separate passes, counters only — MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC slots; on
    gfx950 it tallies the 128-byte requests of a wide streaming read at 64 bytes: x 2; both in KiB).  -> (GB per launch,
    source) or (None, None) when rocprofv3 is not on the box or a pass fails (the caller falls back to the committed profile)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or (os.path.exists("/opt/rocm/bin/rocprofv3") and "/opt/rocm/bin/rocprofv3")
    if not prof:
        return None, None
    vals = {}
    tmp = tempfile.mkdtemp(prefix="fz_traffic_", dir="/tmp")
    try:
        for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, cnt)
            env = dict(os.environ, TMPDIR="/tmp")
            cmd = [prof, "--pmc", cnt, "--output-format", "csv", "-d", out_dir, "--", sys.executable, os.path.abspath(__file__),
                   "--traffic-child", "--mib", str(mib)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
            if r.returncode != 0:
                return None, None
            got = []
            for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if "fz_scan_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == cnt:
                            got.append(float(row["Counter_Value"]))
            if len(got) < 4:
                return None, None
            vals[cnt] = float(np.mean(got[2:])) * 1024.0             # KiB -> bytes; the first launches are the cold ones
        traffic = 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]
        return round(traffic / 1e9, 4), "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around two child runs of the workload"
    except Exception:  # noqa: BLE001 — a profiler hiccup must not cost the run its line
        return None, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


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


class stdout_to_stderr(object):
    """RCCL prints a version banner on the process's C stdout when a communicator is created; this line-oriented
    program owes its stdout ONE JSON line.  Inside the block file descriptor 1 points at stderr (C stdio flushed on both
    sides)."""

    def __enter__(self):
        import ctypes
        self._libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def pipelined_steps(engine, handle, p, k, steps, per_dev=None, gather=None):
    """`steps` searches with two in flight (fz_lev_ngrams_begin / _end): every one starts and completes inside the
    call and delivers its ordered stream.  -> (seconds, last stream)."""
    t0 = time.perf_counter()
    engine.lev_ngrams_begin(handle, p, k)
    matches = None
    for i in range(steps):
        if i + 1 < steps:
            engine.lev_ngrams_begin(handle, p, k)
        matches = engine.lev_ngrams_end(as_array=True)
        if per_dev is not None:
            per_dev.append(engine.device_ms())
        if gather is not None:
            gather.append(engine.comm_gather_ms())
    return time.perf_counter() - t0, matches


def main_multi_device(args):
    """--gpus N launched as ONE process (no torchrun, no torch): a torch-free multi-device context.  BASELINE
    configs[4]: N shards of --mib MiB (4 GiB) each, built shard by shard (never a 4N GiB host array), one per device
    of the context (fz_seq_new / fz_seq_add_shard), (m + k)-byte halos, hits owned by index, copies of the pattern
    planted around every shard boundary and asserted in the merged stream.  A step = one search over ALL shards, two
    searches in flight.

    N distinct devices (the driver's `bench.py --gpus N`): the context joins an RCCL communicator (ncclCommInitAll, one
    rank per device) and the timed search is the COLLECTIVE one — every device scans its shard, snapshots its counters +
    records on the device, ONE ncclAllGather per search over xGMI, every rank's block lands on the host, per-device
    worker threads order the ranks' records, the caller merges them (`rccl_ranks`, `allgather_ms`); the same run also
    times the form without a collective (`value_no_collective`: per-device record lists written to pinned host memory
    and merged on the host) and ONE shard of the same size searched alone on the first device (`scaling_ref_1gpu`), so
    that "x at N vs 1" compares like with like.
    FZ_DEVICES="0,0" maps the N device states onto the listed devices (several states on one GPU: a functional check
    of the code path, not a scaling number; RCCL needs one rank per GPU, so duplicates run without the collective)."""
    N = args.gpus
    env = os.environ.get("FZ_DEVICES", "").strip()
    devices = [int(x) for x in env.split(",") if x.strip()] if env else list(range(N))
    if len(devices) != N:
        raise SystemExit("bench.py: FZ_DEVICES lists %d devices but --gpus is %d" % (len(devices), N))
    from tests import workloads
    k = 2
    if args.mib <= 0:
        args.mib = 4096
    shard_bytes = args.mib << 20
    pattern = workloads.dna(20, 1)
    m = len(pattern)
    p = pattern.tobytes()
    global_n = shard_bytes * N

    cpu = None
    if not args.no_cpu_baseline:                                 # forks: before the HIP runtime exists in this process
        sample, _pat, _pl = workloads.cfg2(min(shard_bytes, 1 << 30), 1024)
        cpu = cpu_baseline(sample, pattern, k, args.cpu_sample_mib)
        cpu["sample"] += " (the first GiB of a shard's workload)"
        cpu_rows, cpu_rows_bytes = cpu.pop("_rows"), cpu.pop("_rows_bytes")

    from fuzzysearch_amd import _native
    lib = _native.load_library()
    import ctypes
    have = ctypes.c_int(0)
    _native._check(lib.fz_device_count(ctypes.byref(have)))
    if max(devices) >= have.value:
        raise SystemExit("bench.py: --gpus %d needs devices %r but only %d HIP device(s) are visible "
                         "(launch under torch.distributed.run for one rank per GPU, or set FZ_DEVICES)" % (N, devices, have.value))
    distinct = len(set(devices)) == len(devices)
    # Pre-flight: what a first run on a new node can trip over is said in ONE sentence, not as a stack trace half an hour in.
    probe = _native.Engine(sorted(set(devices)))
    free_b, total_b = probe.mem_info()                           # the smallest free / total device memory over the devices
    del probe
    per_dev_states = max(devices.count(dv) for dv in set(devices))
    need_b = max((devices.count(dv) + (1 if dv == devices[0] else 0)) * (shard_bytes + (64 << 20)) for dv in set(devices))
    if free_b < need_b:
        raise SystemExit("bench.py: --gpus %d --mib %d needs %.1f GiB of free device memory per GPU (%d shard(s) + the 1-GPU reference "
                         "shard + result buffers) but only %.1f of %.1f GiB are free; use a smaller --mib"
                         % (N, args.mib, need_b / 2.0 ** 30, per_dev_states, free_b / 2.0 ** 30, total_b / 2.0 ** 30))
    # several device states on ONE GPU can only join a communicator of the test suite's stand-in library
    # (tests/mock_rccl.cpp through FZ_RCCL_LIB: tests/test_gpu_mock_rccl.py); RCCL needs one rank per GPU
    stand_in = bool(os.environ.get("FZ_RCCL_LIB")) and _native.Engine.comm_backend() == "stand-in"
    collective = (distinct or stand_in) and os.environ.get("FZ_BENCH_NO_COLLECTIVE") != "1"
    engine = _native.Engine(devices)
    ref_engine = _native.Engine([devices[0]])                    # one shard alone on the first device: the like-for-like 1-GPU figure
    if cpu is not None:
        # the CPU leg and the GPU agree row by row on the CPU leg's sample (ordered raw streams), before anything is timed
        hs = ref_engine.upload(sample[:cpu_rows_bytes])
        gpu_rows = [tuple(int(x) for x in r)[:3] for r in ref_engine.lev_ngrams(hs, p, k, as_array=True).tolist()]
        assert gpu_rows == cpu_rows, "the CPU baseline and the GPU returned different raw streams on the CPU leg's sample"
        cpu["rows_equal_gpu"] = len(cpu_rows)
        hs.release()
        del sample, cpu_rows, gpu_rows
    t_build = time.perf_counter()
    fill, edge_plants = workloads.cfg5_fill(shard_bytes, N, pattern, k)
    handle = engine.new_sequence(global_n)
    ref_handle = None
    for r, buf, off, lo, hi in workloads.iter_shard_buffers(N, shard_bytes, m + k, fill):
        engine.add_shard(handle, r, buf, off, lo, hi)
        if r == 0:
            ref_handle = ref_engine.upload(buf[:shard_bytes])
    t_build = time.perf_counter() - t_build
    rccl_ranks = 0
    collective_error = None
    if collective:
        try:
            with stdout_to_stderr():
                engine.comm_init_all()                           # ncclCommInitAll: every device of the context = one rank
                first_coll = engine.lev_ngrams(handle, p, k, as_array=True)      # (RCCL sets its channels up on first use)
            rccl_ranks = engine.comm_info()[0]
        except Exception as exc:  # noqa: BLE001
            # a node whose RCCL cannot set this communicator up (no peer access, a missing library) still gets its scaling
            # number: the same shards searched without the collective, per-device record lists merged on the host — and the
            # line says so (`rccl_ranks` 0, `collective_error`)
            collective_error = "%s: %s" % (type(exc).__name__, exc)
            sys.stderr.write("bench.py: the collective form is not available (%s); running the host-merged form\n" % collective_error)
            try:
                engine.comm_destroy()
            except Exception:  # noqa: BLE001
                pass
            collective = False

    def timed_phase(collective_now):
        """Settle (adaptive, as main()), warm-up, exactly --steps timed searches, 20 synchronous ones."""
        first_ = engine.lev_ngrams(handle, p, k, as_array=True)
        t_settle = time.perf_counter()
        spans, settled = [], False
        while True:
            now_ms = (time.perf_counter() - t_settle) * 1e3
            if now_ms >= args.settle_ms or (settled and now_ms >= args.settle_min_ms):
                break
            assert np.array_equal(engine.lev_ngrams(handle, p, k, as_array=True), first_), "non-deterministic result"
            spans.append(float(np.mean(engine.device_ms())))
            if len(spans) >= 64 and len(spans) % 16 == 0:
                a_, b_ = float(np.median(spans[-32:])), float(np.median(spans[-64:-32]))
                settled = b_ > 0 and abs(a_ - b_) / b_ < 0.005
        settle_ = {"ms": round((time.perf_counter() - t_settle) * 1e3, 1), "searches": len(spans) + 1, "converged": bool(settled)}
        for _ in range(args.warmup):
            engine.lev_ngrams(handle, p, k, as_array=True)
        per_dev_, gather_ = [], ([] if collective_now else None)
        if args.sync:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                matches_ = engine.lev_ngrams(handle, p, k, as_array=True)
                per_dev_.append(engine.device_ms())
                if gather_ is not None:
                    gather_.append(engine.comm_gather_ms())
            elapsed_ = time.perf_counter() - t0
        else:
            elapsed_, matches_ = pipelined_steps(engine, handle, p, k, args.steps, per_dev_, gather_)
        st_ = engine.stats()
        assert np.array_equal(matches_, first_), "pipelined and synchronous searches returned different streams"
        t1 = time.perf_counter()
        for _ in range(20):
            engine.lev_ngrams(handle, p, k, as_array=True)
        return first_, matches_, elapsed_, per_dev_, gather_, st_, (time.perf_counter() - t1) / 20 * 1e3, settle_

    try:
        first, matches, elapsed, per_dev, gather, st, sync_ms, settle = timed_phase(collective)
    except Exception as exc:  # noqa: BLE001
        if not collective:
            raise
        # a collective failed or ran into its deadline in the middle of the run (a rank that never arrives, an error out of
        # ncclAllGather): the line is still owed — the same shards searched without the collective, and the reason
        collective_error = "%s: %s" % (type(exc).__name__, exc)
        sys.stderr.write("bench.py: the collective search failed (%s); running the host-merged form\n" % collective_error)
        for _ in range(2):                                       # searches that were in flight: collected or dropped
            try:
                engine.lev_ngrams_end(as_array=True)
            except Exception:  # noqa: BLE001
                pass
        engine.comm_set_collective(False)
        collective = False
        rccl_ranks = 0
        first, matches, elapsed, per_dev, gather, st, sync_ms, settle = timed_phase(False)

    # the same search without the collective (per-device records merged on the host), same run
    no_coll = None
    if collective:
        engine.comm_set_collective(False)
        for _ in range(max(3, args.warmup // 4)):
            local = engine.lev_ngrams(handle, p, k, as_array=True)
        assert np.array_equal(local, first), "collective and host-merged searches returned different streams"
        dt, local = pipelined_steps(engine, handle, p, k, args.steps)
        assert np.array_equal(local, first)
        no_coll = {"value": round(global_n * args.steps / dt / 1e9, 2), "ms_per_step": round(dt / args.steps * 1e3, 4)}
        engine.comm_set_collective(True)
    # one shard of the same size alone on the first device, two in flight: the like-for-like 1-GPU reference
    ref_first = ref_engine.lev_ngrams(ref_handle, p, k, as_array=True)
    for _ in range(max(3, args.warmup // 4)):
        ref_engine.lev_ngrams(ref_handle, p, k, as_array=True)
    ref_dev = []
    ref_dt, ref_last = pipelined_steps(ref_engine, ref_handle, p, k, args.steps, ref_dev)
    assert np.array_equal(ref_last, ref_first)
    ref_value = shard_bytes * args.steps / ref_dt / 1e9

    rows = [tuple(int(x) for x in r) for r in matches.tolist()]
    keys = [(g, s_) for (s_, e_, d_, g) in rows]
    found = {(s_, e_, d_) for (s_, e_, d_, _g) in rows}
    missing = [q for q in edge_plants if (q, q + m, 0) not in found]
    assert not missing, "matches across shard boundaries are missing: %r" % (missing[:8],)
    import fuzzysearch_amd as fa
    consolidated = fa.common._native.consolidate(rows)
    per_dev = np.asarray(per_dev)                                # steps x devices
    f_ms = float(per_dev.mean())
    achieved = shard_bytes / (f_ms * 1e-3) / 1e9
    value = global_n * args.steps / elapsed / 1e9
    out = {
        "metric": "GB/s of sequence scanned at |p|=20 max_l_dist=2 (levenshtein_ngram path%s)"
                  % ("" if args.sync else "; two searches in flight"),
        "value": round(value, 2),
        "unit": "GB/s",
        "n_gpus": N,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "%d MiB iid random DNA bytes per GPU, |pattern|=20, max_l_dist=2, 1024 planted variants per GiB "
                               "(BASELINE configs[4]: %d GiB over %d GPUs, copies of the pattern around every shard boundary); "
                               "resident in HBM" % (args.mib, args.mib * N >> 10, N),
                   "bytes_per_gpu": shard_bytes, "pattern_len": m, "max_l_dist": k,
                   "calls_in_flight": 1 if args.sync else 2,
                   "devices": devices,
                   "sharding": ("one process, one device state per GPU (torch-free fz_ctx): contiguous shards, (m+k)-byte halo, "
                                "hits owned by index; " +
                                ("every device = one RCCL rank (ncclCommInitAll): device snapshot of counters + records, ONE "
                                 "ncclAllGather of the ranks' record lists per search, every rank's block read on the host; the "
                                 "all-gather of step i runs next to the scan of step i+1" if collective else
                                 "per-device record lists merged on the host; no collective (%s)"
                                 % ("the devices listed are not distinct: RCCL needs one rank per GPU" if not (distinct or stand_in)
                                    else "the communicator could not be set up: see collective_error" if collective_error
                                    else "FZ_BENCH_NO_COLLECTIVE=1")))},
        "rccl_ranks": rccl_ranks,
        "collective_library": _native.Engine.comm_backend() if collective else None,
        "collective_error": collective_error,
        "settle": settle,
        "allgather_ms": None if not gather else round(float(np.mean(gather)), 4),
        "value_no_collective": round(value, 2) if no_coll is None else no_coll["value"],
        "no_collective_ms_per_step": (round(elapsed / args.steps * 1e3, 4) if no_coll is None else no_coll["ms_per_step"]),
        "scaling_ref_1gpu": {"value": round(ref_value, 2), "unit": "GB/s", "ms_per_step": round(ref_dt / args.steps * 1e3, 4),
                             "scan_kernel_ms": round(float(np.mean(ref_dev)), 4),
                             "x_vs_1gpu": round(value / ref_value, 3),
                             "note": "shard 0 of this run (%d MiB, same bytes) searched alone on device %d by a single-device "
                                     "context, two searches in flight, same process and run" % (args.mib, devices[0])},
        "matches_per_s": round(len(rows) * args.steps / elapsed, 1),
        "raw_matches": len(rows),
        "consolidated_matches": len(consolidated),
        "boundary_plants_found": len(edge_plants),
        "stream_in_reference_order": keys == sorted(keys),
        "ngram_hits": st["ngram_hits"],
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": "fz_scan_kernel", "avg_kernel_ms": round(f_ms, 4),
                     "algorithmic_bytes_per_launch": shard_bytes,
                     "note": "per GPU: one launch scans one shard; average over devices and steps"},
        "kernel_ms": {"filter_per_device": [round(float(x), 4) for x in per_dev.mean(axis=0)]},
        "sync_ms_per_call": round(sync_ms, 4),
        "host_threads": "one worker per device" if (N > 1 and os.environ.get("FZ_NO_DEV_THREADS") is None) else "calling thread only",
        "build_s": round(t_build, 2),
        "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)                                                # whatever the libraries print while shutting down is not stdout's business


_line_printed = [False]
_abandoned_thread = [False]


def _with_deadline(fn, seconds, what):
    """fn() on a helper thread; TimeoutError if it has not returned after `seconds` (a blocking call into a collective
    library that waits for a rank that never comes cannot be interrupted: the thread is abandoned, the process leaves
    through os._exit at its end)."""
    import threading
    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as exc:  # noqa: BLE001 — handed to the caller
            box["error"] = exc
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        _abandoned_thread[0] = True
        raise TimeoutError("%s did not return within %d s (FZ_COMM_INIT_TIMEOUT_S)" % (what, seconds))
    if "error" in box:
        raise box["error"]
    return box.get("value")


def main():
    args = parse()
    if args.traffic_child:
        return traffic_child(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and os.environ.get("FZ_BENCH_FORCE_DIST") != "1" and (args.gpus > 1 or os.environ.get("FZ_BENCH_FORCE_COLLECTIVE") == "1"):
        # FZ_BENCH_FORCE_COLLECTIVE=1: the N-device code path (RCCL communicator over the context's devices) with N = 1 too
        return main_multi_device(args)
    if world == 1 or os.environ.get("FZ_BENCH_TORCH") == "1":
        return main_rank(args)
    # One process per GPU under a launcher (the bench contract's N > 1 form), RCCL behind the C-ABI.  Whatever the collective
    # library does on a node nobody has run this on — refuse the communicator, fail an all-gather in the middle of the run, wait
    # for a rank for ever (every wait has a deadline) — rank 0 still prints ITS LINE: main_launcher_fallback.
    try:
        rc = main_rank(args)
        if _abandoned_thread[0]:
            sys.stdout.flush()
            os._exit(0)
        return rc
    except AssertionError:
        raise                                                    # a wrong result is not a collective problem
    except Exception as exc:  # noqa: BLE001
        if _line_printed[0]:
            raise
        import traceback
        traceback.print_exc(file=sys.stderr)
        err = "%s: %s" % (type(exc).__name__, exc)
    rc = main_launcher_fallback(args, err)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0 if not rc else int(rc))                           # (an abandoned thread inside the collective library must not hold the exit)


def main_launcher_fallback(args, err):
    """The launcher form without a collective library: every rank searches its shard (same workload, same shards, same
    K steps with two searches in flight), barriers and the hand-over of the ranks' streams go through files
    (fuzzysearch_amd.distributed.FileRendezvous), rank 0 merges the streams of the last step ONCE, checks them (boundary
    plants, reference order) and prints the line: `collective_error` says why, `rccl_ranks` is 0, `value` is the aggregate
    of the per-rank searches — the path's one exchange step (the all-gather of the record lists) is NOT inside the timed
    region of this line, and the line says that too."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from fuzzysearch_amd import _native
    from fuzzysearch_amd import distributed as fzd
    from tests import workloads
    comm_ms = float(os.environ.get("FZ_COMM_TIMEOUT_MS", "60000"))
    rv = fzd.FileRendezvous(world, rank, timeout=max(240.0, 3.0 * comm_ms / 1e3 + 120.0))
    errs = [b.decode("utf-8", "replace") for b in rv.allgather(err.encode())]      # every rank has given up on the collective form
    if rank == 0:
        sys.stderr.write("bench.py: the collective form failed (%s); running the per-rank form, streams merged through files\n" % errs[0])
    k = 2
    if args.mib <= 0:
        args.mib = 4096
    shard_bytes = args.mib << 20
    pattern = workloads.dna(20, 1)
    m = len(pattern)
    halo = m + k
    p = pattern.tobytes()
    global_n = shard_bytes * world
    seq = np.empty(shard_bytes, dtype=np.uint8)
    fill, edge_plants = workloads.cfg5_fill(shard_bytes, world, pattern, k)
    fill(rank, seq)
    head, tail = seq[:halo], seq[-halo:]
    edges = [(np.frombuffer(b[:len(b) // 2], dtype=np.uint8), np.frombuffer(b[len(b) // 2:], dtype=np.uint8))
             for b in rv.allgather(head.tobytes() + tail.tobytes())]
    left, right = fzd.halos_from_edges(edges, rank, halo)
    engine = _native.Engine([fzd.local_device(local_rank)])
    buf = np.concatenate([left, seq, right])
    own_lo = rank * shard_bytes
    handle = engine.upload_shard(buf, own_lo - len(left), own_lo, own_lo + shard_bytes, global_n)
    del buf, seq
    first = engine.lev_ngrams(handle, p, k, as_array=True)
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < max(args.settle_min_ms, 300.0):
        assert np.array_equal(engine.lev_ngrams(handle, p, k, as_array=True), first), "non-deterministic result"
    for _ in range(args.warmup):
        engine.lev_ngrams(handle, p, k, as_array=True)
    rv.barrier()                                                 # (file barrier: the ranks start within about a millisecond of each other)
    filter_ms = []
    t0 = time.perf_counter()
    engine.lev_ngrams_begin(handle, p, k)
    last = None
    for i in range(args.steps):
        if i + 1 < args.steps:
            engine.lev_ngrams_begin(handle, p, k)
        last = engine.lev_ngrams_end(as_array=True)
        filter_ms.append(engine.kernel_ms()[0])
    elapsed = time.perf_counter() - t0
    st = engine.stats()
    assert np.array_equal(last, first), "non-deterministic result"
    rows = np.ascontiguousarray(np.asarray([tuple(int(x) for x in r) for r in last.tolist()], dtype=np.int64).reshape(-1, 4))
    head_blob = np.asarray([elapsed, float(np.mean(filter_ms)), float(st["ngram_hits"])], dtype=np.float64).tobytes()
    blobs = rv.allgather(head_blob + rows.tobytes())
    if rank == 0:
        import fuzzysearch_amd as fa
        heads = [np.frombuffer(b[:24], dtype=np.float64) for b in blobs]
        streams = [np.frombuffer(b[24:], dtype=np.int64).reshape(-1, 4) for b in blobs]
        merged = fzd.merge_rank_streams(streams)
        matches = [tuple(int(x) for x in r) for r in merged.tolist()]
        found = {(s_, e_, d_) for (s_, e_, d_, _g) in matches}
        missing = [q for q in edge_plants if (q, q + m, 0) not in found]
        assert not missing, "matches across shard boundaries are missing: %r" % (missing[:8],)
        elapsed_max = max(float(h[0]) for h in heads)
        f_ms = float(np.mean([h[1] for h in heads]))
        value = global_n * args.steps / elapsed_max / 1e9
        achieved = shard_bytes / (f_ms * 1e-3) / 1e9
        out = {
            "metric": "GB/s of sequence scanned at |p|=20 max_l_dist=2 (levenshtein_ngram path; two searches in flight; PER-RANK FORM: the "
                      "collective library failed, see collective_error)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d MiB iid random DNA bytes per GPU, |pattern|=20, max_l_dist=2, 1024 planted variants per GiB "
                                   "(BASELINE configs[4]: %g GiB over %d GPUs, copies of the pattern around every shard boundary); resident in HBM"
                                   % (args.mib, args.mib * world / 1024.0, world),
                       "bytes_per_gpu": shard_bytes, "pattern_len": m, "max_l_dist": k, "calls_in_flight": 2,
                       "sharding": "one process per GPU; contiguous shards, (m+k)-byte halo, hits owned by index; NO collective in this line: "
                                   "every rank keeps its own ordered stream, the streams of the last step are merged once through files "
                                   "(outside the timed region) and checked"},
            "rccl_ranks": 0, "collective_library": None, "collective_error": "; ".join(sorted(set(errs))),
            "exchange_in_timed_region": False, "barrier": "file rendezvous in /dev/shm (the ranks start within about a millisecond)",
            "allgather_ms": None, "value_no_collective": round(value, 2),
            "no_collective_ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "per_rank_ms_per_step": [round(float(h[0]) / args.steps * 1e3, 4) for h in heads],
            "matches_per_s": round(len(matches) * args.steps / elapsed_max, 1), "raw_matches": len(matches),
            "consolidated_matches": len(fa.common._native.consolidate(matches)),
            "boundary_plants_found": len(edge_plants),
            "stream_in_reference_order": [(g_, s_) for (s_, e_, d_, g_) in matches] == sorted((g_, s_) for (s_, e_, d_, g_) in matches),
            "ngram_hits": int(sum(h[2] for h in heads)),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "fz_scan_kernel",
                         "avg_kernel_ms": round(f_ms, 4), "algorithmic_bytes_per_launch": shard_bytes,
                         "note": "per GPU: one launch scans one shard; mean hipEvent span over ranks and steps (two searches in flight)"},
            "csrc_digest": _csrc_digest(), "cpu_baseline": None,
        }
        print(json.dumps(out), flush=True)
        _line_printed[0] = True
    handle.release()
    rv.close()
    return 0


def main_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    # FZ_BENCH_FORCE_DIST=1: take the multi-rank code path with a single rank too (1-GPU smoke of it)
    use_dist = world > 1 or os.environ.get("FZ_BENCH_FORCE_DIST") == "1"
    # N > 1 ranks: RCCL behind the C-ABI (fz_comm_*; the unique id travels through a rendezvous file) — no torch in
    # this process.  FZ_BENCH_TORCH=1 takes the torch.distributed glue instead (same data path, torch collectives).
    use_torch = use_dist and os.environ.get("FZ_BENCH_TORCH") == "1"
    if use_torch:
        # import torch BEFORE libfzhip so both share one HIP runtime (same libamdhip64 SONAME)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from fuzzysearch_amd import _native
    from fuzzysearch_amd import distributed as fzd
    from tests import workloads
    if use_torch:
        from tests import torch_glue          # (test / bench infrastructure: the product never imports torch)

    k = 2
    if args.mib <= 0:
        args.mib = 1024 if world == 1 else 4096
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
        # BASELINE configs[4]: every rank generates its own shard (seed + rank, 1 GiB pieces), 1024 planted
        # variants per GiB, plus exact copies around every shard boundary at deltas {-m-k ... +1}
        # (tests/workloads.py::cfg5_fill; every rank writes the bytes that fall into its shard)
        seq = np.empty(shard_bytes, dtype=np.uint8)
        fill, edge_plants = workloads.cfg5_fill(shard_bytes, world, pattern, k)
        fill(rank, seq)

    cpu = None
    if not args.no_cpu_baseline and rank == 0 and not use_torch:
        # forks: before the HIP runtime exists.  N > 1: rank 0 times the first GiB of its shard while the other ranks
        # wait for the communicator's unique id (the timed region starts behind a barrier)
        cpu = cpu_baseline(seq[:1 << 30] if world > 1 else seq, pattern, k, args.cpu_sample_mib)
    cpu_rows = cpu.pop("_rows") if cpu else None
    cpu_rows_bytes = cpu.pop("_rows_bytes") if cpu else 0

    engine = _native.Engine([fzd.local_device(local_rank) if use_dist else local_rank])
    if use_dist and not use_torch:
        with stdout_to_stderr():
            # joins the job's RCCL communicator: searches are collective from here on (ncclCommInitRank blocks until every
            # rank has called it: under a deadline)
            _with_deadline(lambda: fzd.init_engine_from_env(engine), int(os.environ.get("FZ_COMM_INIT_TIMEOUT_S", "300")),
                           "joining the RCCL communicator")
    if not use_dist:
        handle = engine.upload(seq)
    else:
        # halo exchange, once at load: neighbours' (m + k) edge bytes (one all-gather)
        left, right = torch_glue.exchange_halos(seq, halo) if use_torch else fzd.exchange_halos_native(engine, seq, halo)
        buf = np.concatenate([left, seq, right])
        own_lo = rank * shard_bytes
        handle = engine.upload_shard(buf, own_lo - len(left), own_lo, own_lo + shard_bytes, global_n)
        del buf

    def sync():
        if use_torch:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        elif use_dist:
            engine.comm_barrier()                 # waits for this rank's streams, then an all-reduce over the ranks

    def finish(raw):
        # torch glue: this rank's stream -> the merged global stream (ONE all_gather: counts + packed records);
        # native: the search itself was collective and `raw` already is the merged stream
        return torch_glue.allgather_matches(raw, as_array=True) if use_torch else raw

    def step():
        # synchronous: on return the ordered raw match stream is on the host (numpy view of the
        # C-ABI's fz_match array; no per-record Python objects inside the timed region)
        return finish(engine.lev_ngrams(handle, p, k, as_array=True))

    if cpu_rows is not None and not use_dist and cpu_rows_bytes == len(seq):
        # the CPU leg (the reference's natives on the whole input) and the GPU agree row by row — ordered raw streams,
        # not just counts — before anything is timed
        gpu_rows = [tuple(int(x) for x in r)[:3] for r in engine.lev_ngrams(handle, p, k, as_array=True).tolist()]
        assert gpu_rows == cpu_rows, "the CPU baseline and the GPU returned different raw streams"
        cpu["rows_equal_gpu"] = len(cpu_rows)
        del gpu_rows
    cpu_rows = None
    # setup self-check + clock settle (untimed): repeated searches must return the identical stream
    def any_rank(flag):
        # a time-based loop around COLLECTIVE searches must take the same number of trips on every rank (a rank that
        # leaves early would meet the others' all-gather with its next collective: found by the N-rank tests on the
        # stand-in library, tests/test_gpu_mock_rccl.py) — the ranks agree on every trip
        if use_torch:
            f = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(f, op=dist.ReduceOp.MAX)
            return bool(f.item() > 0)
        return engine.comm_max(1.0 if flag else 0.0) > 0 if use_dist else flag

    # Adaptive settle (untimed): searches until the kernel's span has stopped moving — the median of the last 32 spans within
    # 0.5 % of the median of the 32 before them — or --settle-ms (1 500) is up; at least --settle-min-ms.  A fresh lease ramps
    # its clocks over hundreds of milliseconds, a step is ~0.3 ms: a fixed 150 ms left the timed region on the ramp on some
    # boxes (round 5: the driver's kernel span 0.214 ms against 0.201 in the same run's synchronous block).
    t_settle = time.perf_counter()
    first = step()
    spans = []
    settled = False

    def settle_more():
        now_ms = (time.perf_counter() - t_settle) * 1e3
        if now_ms >= args.settle_ms:
            return False
        return now_ms < args.settle_min_ms or not settled

    while any_rank(settle_more()):
        again = step()
        assert np.array_equal(again, first), "non-deterministic result: two searches returned different streams"
        spans.append(engine.kernel_ms()[0])
        if len(spans) >= 64 and len(spans) % 16 == 0:
            a_, b_ = float(np.median(spans[-32:])), float(np.median(spans[-64:-32]))
            settled = b_ > 0 and abs(a_ - b_) / b_ < 0.005
    settle = {"ms": round((time.perf_counter() - t_settle) * 1e3, 1), "searches": len(spans) + 1, "converged": bool(settled),
              "rule": "median of the last 32 kernel spans within 0.5 % of the 32 before; cap --settle-ms"}
    for _ in range(args.warmup):
        matches = step()
    filter_ms, verify_ms, device_ms, gather_ms = [], [], [], []
    native_dist = use_dist and not use_torch
    sync()
    t0 = time.perf_counter()
    # Two-deep pipeline through the C-ABI (fz_lev_ngrams_begin / _end, two pinned result slots per device):
    # the scan of step i + 1 is on the GPU while the host orders and consumes the records of step i (and,
    # for N > 1, while RCCL all-gathers them: the scan runs on the engine's own HIP stream, the collective
    # and its staging copies on torch's).  Every step still ends with its complete, ordered match list
    # on this rank's host, and exactly K searches (and K all-gathers) start and complete inside the timed
    # region.  args.sync times plain synchronous fz_lev_ngrams calls instead.
    if args.sync and not use_dist:
        for _ in range(args.steps):
            matches = step()
            f_, v_, d_ = engine.kernel_ms()                # hipEvent spans of this step's kernels
            filter_ms.append(f_)
            verify_ms.append(v_)
            device_ms.append(d_)
    else:
        engine.lev_ngrams_begin(handle, p, k)
        for i in range(args.steps):
            if i + 1 < args.steps:
                engine.lev_ngrams_begin(handle, p, k)
            raw = engine.lev_ngrams_end(as_array=True)
            f_, v_, d_ = engine.kernel_ms()
            filter_ms.append(f_)
            verify_ms.append(v_)
            device_ms.append(d_)
            if native_dist:
                gather_ms.append(engine.comm_gather_ms())
            matches = finish(raw)
    sync()
    elapsed = time.perf_counter() - t0
    if use_torch:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    elif use_dist:
        elapsed = engine.comm_max(elapsed)

    st = engine.stats()
    no_coll = ref1 = None
    if native_dist:
        # the same step without the exchange: every rank searches its shard and keeps its own stream (same run)
        engine.comm_set_collective(False)
        for _ in range(max(3, args.warmup // 4)):
            engine.lev_ngrams(handle, p, k, as_array=True)
        sync()
        dt, _local = pipelined_steps(engine, handle, p, k, args.steps)
        sync()
        dt = engine.comm_max(dt)
        no_coll = {"value": round(global_n * args.steps / dt / 1e9, 2), "ms_per_step": round(dt / args.steps * 1e3, 4)}
        # ... and rank 0's shard searched while the other GPUs are idle: the like-for-like 1-GPU figure
        if rank == 0:
            ref_dev = []
            dt1, _l = pipelined_steps(engine, handle, p, k, args.steps, None)
            ref1 = {"value": round(shard_bytes * args.steps / dt1 / 1e9, 2), "unit": "GB/s", "ms_per_step": round(dt1 / args.steps * 1e3, 4),
                    "note": "rank 0's shard (%d MiB) searched alone (the other ranks wait at a barrier), two searches in flight, "
                            "same process and run" % args.mib}
        sync()
        engine.comm_set_collective(True)
    sync_ms = None
    sync_spans = []
    if not use_dist and not args.sync:
        # latency of one synchronous call (one search in flight), outside the timed region
        for _ in range(10):
            assert np.array_equal(step(), matches), "pipelined and synchronous searches returned different streams"
        batches = []                                   # median of five batches of 40 calls: one descheduled batch is not the latency
        sync_spans = []                                # ... and the kernel's own hipEvent span of every one of them
        for _b in range(5):
            t1 = time.perf_counter()
            for _ in range(40):
                step()
                sync_spans.append(engine.kernel_ms()[0])
            batches.append((time.perf_counter() - t1) / 40 * 1e3)
        sync_ms = sorted(batches)[2]
    two_streams = None
    if not use_dist and not args.sync and args.two_streams:
        # the same two-deep pipeline with the younger scan on a second stream (fz_set_streams(2)): it starts while the older
        # scan drains.  Reported beside `value`, not as `value`: the scans then overlap, so a kernel's own hipEvent span no
        # longer measures the kernel alone (the roofline block above stays on the one-stream loop)
        engine.set_streams(2)
        for _ in range(10):
            assert np.array_equal(step(), matches), "the search returned a different stream after fz_set_streams(2)"
        dt2, last2 = pipelined_steps(engine, handle, p, k, args.steps)
        assert np.array_equal(last2, matches), "two-stream pipeline returned a different stream"
        f2 = engine.kernel_ms()[0]
        engine.set_streams(1)
        two_streams = {"value": round(global_n * args.steps / dt2 / 1e9, 2), "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                       "overlapped_kernel_span_ms": round(f2, 4),
                       "note": "fz_set_streams(2): the younger of the two searches in flight scans on its own stream; same run, same K steps"}
    if rank == 0:
        import fuzzysearch_amd as fa
        matches = [tuple(int(x) for x in r) for r in matches.tolist()]
        if world > 1:
            # every copy planted around a shard boundary is in the merged stream, at its exact position
            found = {(s_, e_, d_) for (s_, e_, d_, _g) in matches}
            missing = [q for q in edge_plants if (q, q + m, 0) not in found]
            assert not missing, "matches across shard boundaries are missing: %r" % (missing[:8],)
        consolidated = fa.common._native.consolidate(matches)
        ms_per_step = elapsed / args.steps * 1e3
        value = global_n * args.steps / elapsed / 1e9
        f_pipe = float(np.mean(filter_ms))                     # spans of the timed region's launches (two searches in flight: a
                                                               # launch queued behind its predecessor overlaps that one's tail)
        f_ms = float(np.mean(sync_spans)) if sync_spans else f_pipe   # spans of 200 synchronous launches: the kernel alone
        achieved = shard_bytes / (f_ms * 1e-3) / 1e9           # algorithmic bytes: N read once
        traffic_gb, traffic_src = (None, None)
        if world == 1 and not use_dist and not args.no_live_traffic:
            traffic_gb, traffic_src = live_traffic(args.mib)
        if traffic_gb is None:
            traffic_gb, traffic_src = measured_traffic(shard_bytes)
        out = {
            "metric": "GB/s of sequence scanned at |p|=20 max_l_dist=2 (levenshtein_ngram path%s)"
                      % ("" if (args.sync and not use_dist) else "; two searches in flight: see value_sync for one synchronous call at a time"),
            "value": round(value, 2),
            "value_sync": None if sync_ms is None else round(global_n / (sync_ms * 1e-3) / 1e9, 2),
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
                                   "1024 planted variants per GiB (BASELINE %s); resident in HBM"
                                   % (args.mib, "configs[1]" if world == 1 else "configs[4]: %d GiB over %d GPUs, copies of the pattern "
                                      "around every shard boundary" % (args.mib * world >> 10, world)),
                       "bytes_per_gpu": shard_bytes, "pattern_len": m, "max_l_dist": k,
                       "calls_in_flight": 1 if (args.sync and not use_dist) else 2,
                       "sharding": "none" if not use_dist else "one process per GPU; contiguous shards, (m+k)-byte halo, hits owned by "
                                   "index; ONE RCCL all-gather of the ranks' record lists per search (%s); the all-gather of "
                                   "step i overlaps the scan of step i+1" % ("torch.distributed glue" if use_torch else
                                                                            "ncclAllGather behind the C-ABI, no torch")},
            "matches_per_s": round(len(matches) * args.steps / elapsed, 1),
            "raw_matches": len(matches),
            "consolidated_matches": len(consolidated),
            "stream_in_reference_order": [(g_, s_) for (s_, e_, d_, g_) in matches] == sorted((g_, s_) for (s_, e_, d_, g_) in matches),
            "ngram_hits": st["ngram_hits"],
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic_gb,
                         "traffic_unit": "GB per launch (rocprofv3 PMC: 2 x FETCH_SIZE [gfx950 correction] + WRITE_SIZE)",
                         "traffic_source": traffic_src,
                         "kernel": "fz_scan_kernel", "avg_kernel_ms": round(f_ms, 4),
                         "avg_kernel_ms_source": ("mean hipEvent span of %d synchronous launches after the timed region" % len(sync_spans))
                                                 if sync_spans else "mean hipEvent span of the timed region's launches",
                         "avg_kernel_ms_sync": round(f_ms, 4) if sync_spans else None,
                         "median_kernel_ms_sync": round(float(np.median(sync_spans)), 4) if sync_spans else None,
                         "min_max_kernel_ms_sync": [round(float(np.min(sync_spans)), 4), round(float(np.max(sync_spans)), 4)] if sync_spans else None,
                         "avg_kernel_ms_pipelined": round(f_pipe, 4),
                         "pipelined_note": "spans of the timed region's launches, two searches in flight: each overlaps its predecessor's "
                                           "tail — not the kernel alone; frac / achieved use the synchronous figure",
                         "algorithmic_bytes_per_launch": shard_bytes},
            "settle": settle,
            "csrc_digest": _csrc_digest(),
            "kernel_ms": {"filter": round(f_pipe, 4), "verify": round(float(np.mean(verify_ms)), 4),
                          "device_total": round(float(np.mean(device_ms)), 4)},
            "sync_ms_per_call": None if sync_ms is None else round(sync_ms, 4),
            "two_streams": two_streams,
        }
        if use_dist:
            out["rccl_ranks"] = world if native_dist else 0
            out["collective_library"] = _native.Engine.comm_backend() if native_dist else "torch.distributed"
            out["allgather_ms"] = round(float(np.mean(gather_ms)), 4) if gather_ms else None
            out["collective_error"] = None                   # (a failing collective library: main_launcher_fallback prints the line)
            out["exchange_in_timed_region"] = True
            if world > 1:
                out["boundary_plants_found"] = len(edge_plants)      # (asserted above: every one is in the merged stream)
            if no_coll:
                out["value_no_collective"] = no_coll["value"]
                out["no_collective_ms_per_step"] = no_coll["ms_per_step"]
            if ref1:
                ref1["x_vs_1gpu"] = round(value / ref1["value"], 3)
                out["scaling_ref_1gpu"] = ref1
        out["cpu_baseline"] = cpu
        if world == 1 and not use_dist and not args.no_extras:
            handle.release()
            del seq
            try:
                out.update(extra_blocks(engine, workloads, max(20, args.steps // 2)))
            except Exception as exc:  # noqa: BLE001 — the headline line is printed whatever happens to the secondary blocks
                out["extras_error"] = "%s: %s" % (type(exc).__name__, exc)
        print(json.dumps(out), flush=True)
        _line_printed[0] = True
    if use_dist:
        _line_printed[0] = True                                  # (every rank: nothing behind this point starts the fallback)
        sys.stdout.flush()
        os.dup2(2, 1)                                            # (RCCL's shutdown messages)
    if use_torch:
        dist.destroy_process_group()
    elif use_dist:
        with stdout_to_stderr():
            try:
                engine.comm_barrier()
                engine.comm_destroy()
            except Exception as exc:  # noqa: BLE001 — the line is out; a failing teardown must not turn the run into an error
                sys.stderr.write("bench.py: communicator teardown: %s: %s\n" % (type(exc).__name__, exc))


if __name__ == "__main__":
    main()
