cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do
echo -n "band        "; FUZZYSEARCH_HIP_LIB=$PWD/benchmarks/lab/libfzhip_base.so python benchmarks/ab_scan.py 1024 300 2>&1 | grep cfg1 | grep -o '"scan_ms.*'
echo -n "bits32 6w   "; FZ_BITS_MIN_K=1 FUZZYSEARCH_HIP_LIB=$PWD/benchmarks/lab/libfzhip_base.so python benchmarks/ab_scan.py 1024 300 2>&1 | grep cfg1 | grep -o '"scan_ms.*'
echo -n "bits32 7w   "; FZ_BITS_MIN_K=1 FUZZYSEARCH_HIP_LIB=$PWD/benchmarks/lab/libfzhip_w47.so python benchmarks/ab_scan.py 1024 300 2>&1 | grep cfg1 | grep -o '"scan_ms.*'
done
