#!/bin/bash
# Lab builds of libfzhip.so for A/B runs on the GPU box: only the headline / exact-search scan kernel
# instances are compiled (-DFZ_LAB_ONLY: seconds instead of minutes), extra -D knobs select kernel variants.
#   benchmarks/lab_build.sh <name> [-DKNOB ...]   ->  benchmarks/lab/libfzhip_<name>.so
# Run with FUZZYSEARCH_HIP_LIB=benchmarks/lab/libfzhip_<name>.so python benchmarks/ab_scan.py
# The product source carries no lab code: the build works on a scratch copy of fuzzysearch_amd/csrc with
# benchmarks/lab_patches/lab_instrumentation.patch applied (time stamps per workgroup phase / n-gram hit: -DFZ_LAB_TIMING,
# -DFZ_LAB_SCANTIME, -DFZ_LAB_LPTIME; kernels with a part of the candidate handling left out: -DFZ_LAB_NOVERIFY / NODP /
# NOEXACT / NOPREFETCH / NOPOOL; -DFZ_LAB_GEN_NO_PRUNE; the miscompiled loop shape -DFZ_LEVLP_BREAKLOOP; -DFZ_LEVLP_STRUCT).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/benchmarks/lab
SCRATCH=$(mktemp -d /tmp/fzlab.XXXXXX)
trap 'rm -rf $SCRATCH' EXIT
mkdir -p $SCRATCH/fuzzysearch_amd $SCRATCH/include
cp -r $ROOT/fuzzysearch_amd/csrc $SCRATCH/fuzzysearch_amd/csrc
cp $ROOT/include/*.h $SCRATCH/include/
(cd $SCRATCH && patch -s -p1 < $ROOT/benchmarks/lab_patches/lab_instrumentation.patch)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-pass-failed \
  "$@" $SCRATCH/fuzzysearch_amd/csrc/fzhip.hip -ldl -Wl,-rpath,/opt/rocm/lib -o $ROOT/benchmarks/lab/libfzhip_$NAME.so
echo built benchmarks/lab/libfzhip_$NAME.so
