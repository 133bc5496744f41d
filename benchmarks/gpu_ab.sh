cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_file_api.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_wf.log
tail -12 gpurun_out/pytest_wf.log
for L in fuzzysearch_amd/libfzhip.so benchmarks/r1/libfzhip_wf_pair1.so; do FUZZYSEARCH_HIP_LIB=$PWD/$L timeout 300 python benchmarks/ab_cfg3a.py; done
