cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest1.log
tail -4 gpurun_out/pytest1.log
(for rep in 1 2; do for L in fuzzysearch_amd/libfzhip.so benchmarks/r1/libfzhip_r1.so; do FUZZYSEARCH_HIP_LIB=$PWD/$L timeout 300 python benchmarks/ab_scan.py 1024 300 --all; done; done) 2>&1 | tee gpurun_out/ab8.log | cut -c1-250
