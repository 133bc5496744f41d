#!/bin/bash
# PMC passes over selected filter_variants kernels (run on the GPU box) -> gpurun_out/pmc_fv/
# NOTE: only the SQ_* pass completed on this pool; the TCP_*/TCC_* passes ran into the 200 s timeout
# (10 GPU-minutes for nothing) — keep them out unless the box is yours for longer.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_fv; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FV_ONLY="stream r4 g8;HV 1 block  ;HV 1 block L2res;H-in-VGPR  ;H-in-VGPR L2-res"
i=0
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_avr TCC_REQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/p$i -- $ROOT/benchmarks/filter_variants 1024 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
find $OUT -name "*counter_collection.csv" | head
