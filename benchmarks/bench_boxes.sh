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
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/boxtrace -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-live-traffic > $OUT/trace_$TAG.log 2>&1
echo "trace rc=$?"
cd $ROOT
python - "$TAG" <<'EOF'
import csv, glob, json, os, subprocess, sys
tag = sys.argv[1]
out = os.path.join("gpurun_out", "boxes")
line = json.loads(open(os.path.join(out, "bench_%s.json" % tag)).read().strip().splitlines()[-1])
trace = max(glob.glob("/tmp/boxtrace/*/*_kernel_trace.csv"), key=os.path.getmtime)
head = "fz_scan_kernel<2, 3, true, false, true, 0>"
grids = {}
for r in csv.DictReader(open(trace)):
    if head in r["Kernel_Name"]:
        g = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0)
        grids.setdefault(g, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
g, durs = max(grids.items(), key=lambda kv: len(kv[1]))          # the 1 GiB launches: the most frequent grid of the run
avg_us = sum(durs) / len(durs) / 1e3
try:
    uid = subprocess.run("rocm-smi --showuniqueid | grep -i 'unique id' | head -1", shell=True, capture_output=True, text=True, timeout=20).stdout.strip().split()[-1]
except Exception:
    uid = "?"
rf = line["roofline"]
sync_ms = rf.get("avg_kernel_ms_sync", rf.get("avg_kernel_ms"))
frac_trace = 2 ** 30 / (avg_us * 1e-6) / 8e12
row = {"box": tag, "gpu_unique_id": uid, "bench_frac": rf["frac"], "bench_avg_kernel_ms_sync": sync_ms, "bench_avg_kernel_ms_pipelined": rf.get("avg_kernel_ms_pipelined"),
       "settle": line.get("settle"), "trace_avg_us": round(avg_us, 2), "trace_calls": len(durs), "trace_frac": round(frac_trace, 4),
       "frac_bench_over_trace": round(rf["frac"] / frac_trace, 4), "value_GBps": line["value"], "value_sync": line.get("value_sync"),
       "traffic": rf.get("traffic"), "traffic_source": rf.get("traffic_source"), "csrc_digest": line.get("csrc_digest")}
open(os.path.join(out, "%s.txt" % tag), "w").write(json.dumps(row) + "\n")
print(json.dumps(row))
EOF
