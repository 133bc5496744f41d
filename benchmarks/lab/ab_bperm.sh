cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for i in 1 2; do for L in base bperm1 bperm2 bperm3 bperm; do FUZZYSEARCH_HIP_LIB=$PWD/benchmarks/lab/libfzhip_$L.so timeout 300 python benchmarks/ab_scan.py 1024 300; done; done 2>&1 | grep -v "^$" | tee gpurun_out/ab_bperm2.txt
