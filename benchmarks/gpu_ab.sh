cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_file_api.py tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_file.log
tail -25 gpurun_out/pytest_file.log
timeout 600 python benchmarks/file_api.py 1024 2>&1 | tee gpurun_out/file_api.log | tail -20
