cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_full.log
tail -6 gpurun_out/pytest_full.log
nproc; free -g | head -2
( time timeout 900 python bench.py ) > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err; tail -3 gpurun_out/bench_r02a.err; cat gpurun_out/bench_r02a.json
