#!/bin/bash
# on the GPU box: benchmarks/trace_pipeline.sh <tag>  -> gpurun_out/pipe_<tag>.txt (gaps between consecutive scan kernels)
TAG=${1:-pipe}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pipe_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $ROOT/benchmarks/trace_pipeline.py > $OUT/run.log 2>&1
python3 - $OUT <<'PY' | tee $OUT/../pipe_$TAG.txt
import csv, glob, os, sys
import statistics as st
out = sys.argv[1]
print(open(os.path.join(out, "run.log")).read().strip().splitlines()[-2:])
rows = []
for f in glob.glob(os.path.join(out, "trace", "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "fz_scan_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))))
rows.sort()
by = {}
for i in range(1, len(rows)):
    if rows[i][2] == rows[i - 1][2]:
        by.setdefault(rows[i][2], []).append((rows[i][0] - rows[i - 1][1], rows[i][1] - rows[i][0]))
for g, v in by.items():
    gaps = [x[0] for x in v]; dur = [x[1] for x in v]
    tail = gaps[-200:]; d = dur[-200:]
    print("grid %d: %d launches; last 200: kernel avg %.1f us, gap median %.1f us, p10 %.1f, p90 %.1f" % (g, len(v), st.mean(d) / 1e3, st.median(tail) / 1e3,
          sorted(tail)[len(tail) // 10] / 1e3, sorted(tail)[len(tail) * 9 // 10] / 1e3))
PY
