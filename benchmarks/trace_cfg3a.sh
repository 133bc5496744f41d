#!/bin/bash
# Kernel trace of benchmarks/ab_cfg3a.py (Levenshtein budgets 5 .. 12 on 1 GiB of text and DNA: the fused and the
# stand-alone lane-per-cell forms as the library chooses them):  benchmarks/trace_cfg3a.sh <tag> -> gpurun_out/trace_<tag>/
TAG=${1:-cfg3a}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python $ROOT/benchmarks/ab_cfg3a.py > $OUT/run.log 2>&1
echo "trace rc=$?"
grep '^{' $OUT/run.log
python3 - $OUT/t <<'PY'
import csv, glob, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*", "*kernel_stats.csv"))):
    for r in csv.DictReader(open(f)):
        print("%-86s calls %6s avg_ns %10s" % (r["Name"][:86], r["Calls"], r["AverageNs"]))
PY
