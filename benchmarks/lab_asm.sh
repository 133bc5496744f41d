#!/bin/bash
# ISA of the headline scan kernel instance for a lab variant:  benchmarks/lab_asm.sh <name> [-DKNOB ...] -> /tmp/asm/<name>.s
# (scratch copy of the sources with benchmarks/lab_patches/lab_instrumentation.patch applied, as benchmarks/lab_build.sh)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p /tmp/asm
SCRATCH=$(mktemp -d /tmp/fzlab.XXXXXX)
trap 'rm -rf $SCRATCH' EXIT
mkdir -p $SCRATCH/fuzzysearch_amd $SCRATCH/include
cp -r $ROOT/fuzzysearch_amd/csrc $SCRATCH/fuzzysearch_amd/csrc
cp $ROOT/include/*.h $SCRATCH/include/
(cd $SCRATCH && patch -s -p1 < $ROOT/benchmarks/lab_patches/lab_instrumentation.patch)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Wno-unused-function -Wno-pass-failed \
  -DFZ_LAB_ONLY "$@" $SCRATCH/fuzzysearch_amd/csrc/fzhip.hip -o /tmp/asm/$NAME.full.s 2>/dev/null
awk '/^_Z14fz_scan_kernelILi2ELi3ELb1ELb0ELb1ELi0EE.*:/{f=1} f{print} /\.end_amdhsa_kernel/{if(f){exit}}' /tmp/asm/$NAME.full.s > /tmp/asm/$NAME.s
python3 - "$NAME" <<'PY'
import re,sys
name=sys.argv[1]
txt=open('/tmp/asm/%s.full.s'%name).read()
for m in re.finditer(r'\.name:\s+(_Z14fz_scan_kernelILi2ELi3ELb1ELb0ELb1ELi0E\S+)\n(.*?)\.wavefront_size', txt, re.S):
    blk=m.group(2)
    print(name, {k:re.search(k+r':\s+(\d+)',blk).group(1) for k in ['.sgpr_count','.vgpr_count','.sgpr_spill_count','.vgpr_spill_count','.private_segment_fixed_size'] if re.search(k+r':\s+(\d+)',blk)})
PY
