cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for i in 1 2 3; do
for L in fuzzysearch_amd/libfzhip.so benchmarks/lab/libfzhip_bp5.so; do FUZZYSEARCH_HIP_LIB=$PWD/$L python benchmarks/ab_scan.py 1024 300 --all 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['lib'][:16].ljust(16), d['workload'][:28].ljust(28), d['ms_per_call'], d['scan_ms'], d['verify_ms'], d['raw'])"; done; done | tee gpurun_out/ab_bp5.txt
