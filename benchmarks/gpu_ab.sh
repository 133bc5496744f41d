set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/pytest1.log
tail -15 gpurun_out/pytest1.log
(timeout 300 python benchmarks/ab_scan.py 1024 200 --all
 FZ_SCAN_FLAGS=1 timeout 300 python benchmarks/ab_scan.py 1024 200
 FZ_WG_PER_CU=8 timeout 300 python benchmarks/ab_scan.py 1024 200
 FZ_WG_PER_CU=16 timeout 300 python benchmarks/ab_scan.py 1024 200
 FZ_SCAN_FLAGS=1 FZ_WG_PER_CU=16 timeout 300 python benchmarks/ab_scan.py 1024 200
 FUZZYSEARCH_HIP_LIB=$PWD/benchmarks/r1/libfzhip_r1.so timeout 300 python benchmarks/ab_scan.py 1024 200) 2>&1 | grep -v "^+" | tee gpurun_out/ab1.log | tail -30
