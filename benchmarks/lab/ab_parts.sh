cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do for L in base qm24 qm16 qm10; do FUZZYSEARCH_HIP_LIB=$PWD/benchmarks/lab/libfzhip_$L.so python benchmarks/ab_scan.py 1024 300 2>&1 | grep cfg1 | grep -o '"lib.*"workload\|"scan_ms.*'| tr '\n' ' '; echo; done; done
