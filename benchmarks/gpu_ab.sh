cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest1.log
tail -8 gpurun_out/pytest1.log
(timeout 300 python benchmarks/ab_scan.py 1024 200 --all
 FUZZYSEARCH_HIP_LIB=$PWD/benchmarks/r1/libfzhip_r1.so timeout 300 python benchmarks/ab_scan.py 1024 200 --all) 2>&1 | tee gpurun_out/ab5.log | cut -c1-250
