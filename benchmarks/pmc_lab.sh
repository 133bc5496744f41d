#!/bin/bash
# PMC passes over ab_scan.py for one lab build (GPU box):  benchmarks/pmc_lab.sh <lib.so> <tag> "<counters pass 1>" "<pass 2>" ...
LIB=$1; TAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CNT in "$@"; do
  i=$((i+1))
  FUZZYSEARCH_HIP_LIB=$ROOT/$LIB timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/p$i -- python $ROOT/benchmarks/ab_scan.py 1024 20 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - $OUT <<'PY'
import collections, csv, glob, sys, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "p*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "fz_scan" not in k: continue
        print(k)
        for c, v in sorted(d.items()):
            print("   %-32s %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
