#!/bin/bash
# One box, one line of profiles/r06_bench_boxes.txt: the default `python bench.py` line's roofline figure (live hipEvent spans of
# synchronous launches) beside the rocprofv3 kernel-trace average of the same kernel on the same box, same code.
# usage (GPU box): bash benchmarks/bench_boxes.sh <tag>   -> gpurun_out/boxes/<tag>.txt (+ the bench line itself)
set -u
TAG=${1:-a}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/boxes
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/boxtrace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/boxtrace -- python $ROOT/bench.py --no-cpu-baseline --no-live-traffic > $OUT/trace_$TAG.log 2>&1
echo "trace rc=$?"
cd $ROOT
python - "$TAG" <<'EOF'
import csv, glob, json, os, subprocess, sys
tag = sys.argv[1]
out = os.path.join("gpurun_out", "boxes")
line = json.loads(open(os.path.join(out, "bench_%s.json" % tag)).read().strip().splitlines()[-1])
trace = max(glob.glob("/tmp/boxtrace/*/*_kernel_trace.csv"), key=os.path.getmtime)
head = "fz_scan_kernel<2, 3, true, false, true, 0>"
rows = []
for r in csv.DictReader(open(trace)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0)))
rows.sort()
grids = {}
for i, (t0, t1, name, g) in enumerate(rows):
    if head in name:
        # a launch with nothing of this process on the device within 5 us before and after it: one of bench.py's synchronous
        # launches (the timed region keeps two searches in flight: those spans overlap their neighbours' tails)
        alone = (i == 0 or t0 - rows[i - 1][1] > 5000) and (i + 1 == len(rows) or rows[i + 1][0] - t1 > 5000)
        grids.setdefault(g, []).append((t1 - t0, alone))
g, spans = max(grids.items(), key=lambda kv: len(kv[1]))          # the 1 GiB launches: the most frequent grid of the run
durs = [d for d, alone in spans if alone]
piped = [d for d, alone in spans if not alone]
avg_all_us = sum(durs) / len(durs) / 1e3
half = durs[len(durs) // 2:]                    # the run's second half: behind the clock ramp of a fresh process (bench.py's figure is taken behind its settle phase)
avg_us = sum(half) / len(half) / 1e3
med_us = sorted(durs)[len(durs) // 2] / 1e3
piped_us = sum(piped) / max(1, len(piped)) / 1e3
try:
    uid = [l.split()[-1] for l in subprocess.run("rocminfo", shell=True, capture_output=True, text=True, timeout=30).stdout.splitlines() if "Uuid" in l and "GPU-" in l][0]
except Exception:
    uid = "?"
rf = line["roofline"]
sync_ms = rf.get("avg_kernel_ms_sync", rf.get("avg_kernel_ms"))
frac_trace = 2 ** 30 / (avg_all_us * 1e-6) / 8e12           # the --stats figure: every launch of the run's headline workload
row = {"box": tag, "gpu_unique_id": uid, "bench_frac": rf["frac"], "bench_avg_kernel_ms_sync": sync_ms, "bench_avg_kernel_ms_pipelined": rf.get("avg_kernel_ms_pipelined"),
       "settle": line.get("settle"), "trace_avg_us_second_half": round(avg_us, 2), "trace_median_us": round(med_us, 2), "trace_avg_us_all": round(avg_all_us, 2), "trace_calls": len(durs), "trace_frac": round(frac_trace, 4),
       "frac_bench_over_trace": round(rf["frac"] / frac_trace, 4), "value_GBps": line["value"], "value_sync": line.get("value_sync"),
       "traffic": rf.get("traffic"), "traffic_source": rf.get("traffic_source"), "csrc_digest": line.get("csrc_digest")}
open(os.path.join(out, "%s.txt" % tag), "w").write(json.dumps(row) + "\n")
print(json.dumps(row))
EOF
