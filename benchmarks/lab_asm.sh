#!/bin/bash
# ISA of the headline scan kernel instance for a lab variant:  benchmarks/lab_asm.sh <name> [-DKNOB ...] -> /tmp/asm/<name>.s
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Wno-unused-function -Wno-pass-failed \
  -DFZ_LAB_ONLY "$@" $ROOT/fuzzysearch_amd/csrc/fzhip.hip -o /tmp/asm/$NAME.full.s 2>/dev/null
awk '/^_Z14fz_scan_kernelILi2ELi3ELb1ELb0ELb1ELi0EE.*:/{f=1} f{print} /\.end_amdhsa_kernel/{if(f){exit}}' /tmp/asm/$NAME.full.s > /tmp/asm/$NAME.s
grep -E "vgpr_count|sgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed|group_segment_fixed" /tmp/asm/$NAME.full.s | grep -A0 "" | awk 'NR<=0'
python3 - "$NAME" <<'PY'
import re,sys
name=sys.argv[1]
txt=open('/tmp/asm/%s.full.s'%name).read()
# metadata block for the kernel
for m in re.finditer(r'\.name:\s+(_Z14fz_scan_kernelILi2ELi3ELb1ELb0ELb1ELi0E\S+)\n(.*?)\.wavefront_size', txt, re.S):
    blk=m.group(2)
    print(name, {k:re.search(k+r':\s+(\d+)',blk).group(1) for k in ['.sgpr_count','.vgpr_count','.sgpr_spill_count','.vgpr_spill_count','.private_segment_fixed_size'] if re.search(k+r':\s+(\d+)',blk)})
PY
