#!/bin/bash
# Kernel trace + PMC passes over benchmarks/regimes.py on the GPU box:  benchmarks/pmc_regimes.sh <tag> "<m,k;m,k>" [mib]
TAG=${1:-reg}; ONLY=${2:-"54,8"}; MIB=${3:-1024}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/benchmarks/regimes.py --only $ONLY --mib $MIB --reps 5"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
echo "trace rc=$?"; grep '^{' $OUT/trace.log
i=0
for CNT in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $CNT --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - $OUT <<'PY'
import collections, csv, glob, sys, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "trace", "*", "*kernel_stats.csv"))):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %6s avg_ns %10s" % (r["Name"][:70], r["Calls"], r["AverageNs"]))
for f in sorted(glob.glob(os.path.join(out, "p*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60] + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "fz_" not in k: continue
        print(k)
        for c, v in sorted(d.items()):
            print("   %-32s %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
