cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do for L in base subh; do FUZZYSEARCH_HIP_LIB=$PWD/benchmarks/lab/libfzhip_$L.so python benchmarks/lab/nohit.py; done; done 2>&1 | grep -v "^$"
