#!/bin/bash
# ISA of the fused bit-vector scan instance (DNA m = 54, k = 8: fz_scan_kernel<2, 3, true, false, true, 1>) -> /tmp/asm/<name>.s
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=${1:-bits}; shift
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Wno-unused-function -Wno-pass-failed \
  -DFZ_LAB_ONLY "$@" $ROOT/fuzzysearch_amd/csrc/fzhip.hip -o /tmp/asm/$NAME.full.s 2>/dev/null
awk -v pat="^_Z14fz_scan_kernelILi2ELi3ELb1ELb0ELb1ELi${WFGV:-1}EE.*:" '$0 ~ pat {f=1} f{print} /\.end_amdhsa_kernel/{if(f){exit}}' /tmp/asm/$NAME.full.s > /tmp/asm/$NAME.s
grep -A40 "amdhsa_kernel _Z14fz_scan_kernelILi2ELi3ELb1ELb0ELb1ELi${WFGV:-1}EE" /tmp/asm/$NAME.full.s | grep -E "next_free_vgpr|next_free_sgpr|private_segment_fixed" 
wc -l /tmp/asm/$NAME.s
