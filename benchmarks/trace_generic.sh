#!/bin/bash
# Kernel trace of configs[3b], raw stream and consolidated (no PMC passes):  benchmarks/trace_generic.sh <tag>
TAG=${1:-gen}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in raw consolidated; do
  if [ $MODE = consolidated ]; then export FZ_AB_CONSOLIDATED=1; else unset FZ_AB_CONSOLIDATED; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$MODE -- python $ROOT/benchmarks/ab_generic.py > $OUT/$MODE.log 2>&1
  echo "== $MODE: trace rc=$?"
  grep '^{' $OUT/$MODE.log
  python3 - $OUT/$MODE <<'PY'
import csv, glob, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*", "*kernel_stats.csv"))):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %6s avg_ns %10s" % (r["Name"][:70], r["Calls"], r["AverageNs"]))
PY
done
