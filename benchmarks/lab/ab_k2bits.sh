cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do python benchmarks/ab_scan.py 1024 300 | head -1; FZ_BITS_MIN_K=1 python benchmarks/ab_scan.py 1024 300 | head -1; FZ_BITS_MIN_K=1 FZ_BITS_QCAP=96 python benchmarks/ab_scan.py 1024 300 | head -1; done 2>&1 | grep -o '"ms_per_call.*'
