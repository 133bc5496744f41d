#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run).
# usage: benchmarks/profile.sh <tag>      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# kernel trace over the whole default bench (headline + 4 GiB target + the other configs); counters over the headline only
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-live-traffic"   # (two searches in flight, as the default bench)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic > $OUT/stats.log 2>&1
echo "stats rc=$?"
i=0
for CNT in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1
  echo "pmc$i rc=$?"
done
find $OUT -name "*.csv" | head -20
