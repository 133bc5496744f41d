#!/bin/bash
# Lab builds of libfzhip.so for A/B runs on the GPU box: only the headline / exact-search scan kernel
# instances are compiled (seconds instead of minutes), extra -D knobs select kernel variants.
#   benchmarks/lab_build.sh <name> [-DKNOB ...]   ->  benchmarks/lab/libfzhip_<name>.so
# Run with FUZZYSEARCH_HIP_LIB=benchmarks/lab/libfzhip_<name>.so python benchmarks/ab_scan.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/benchmarks/lab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-pass-failed \
  "$@" $ROOT/fuzzysearch_amd/csrc/fzhip.hip -ldl -Wl,-rpath,/opt/rocm/lib -o $ROOT/benchmarks/lab/libfzhip_$NAME.so
echo built benchmarks/lab/libfzhip_$NAME.so
