cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do for R in 0 3 2 1; do for L in base qm10; do echo -n "rounds=$R "; FZ_ROUNDS=$R FUZZYSEARCH_HIP_LIB=$PWD/benchmarks/lab/libfzhip_$L.so python benchmarks/ab_scan.py 1024 300 2>&1 | grep cfg1 | grep -o '"lib.*"workload\|"scan_ms.*'| tr '\n' ' '; echo; done; done; done
