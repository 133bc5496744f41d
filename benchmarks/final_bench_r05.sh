mkdir -p gpurun_out/r05
python bench.py > gpurun_out/r05/bench_n1.json 2> gpurun_out/r05/bench_n1.err; echo "n1 rc=$?"
python -c "from tests import mock_rccl; mock_rccl.build()"
FZ_RCCL_LIB=$PWD/tests/libmock_rccl.so FZ_DEVICES=0,0,0,0,0,0,0,0 python bench.py --gpus 8 --mib 1024 --no-cpu-baseline > gpurun_out/r05/bench_8_ranks_stand_in.json 2> gpurun_out/r05/bench_8.err; echo "8 ranks rc=$?"
FZ_DEVICES=0,0 python bench.py --gpus 2 --no-cpu-baseline > gpurun_out/r05/bench_two_device_states.json 2> gpurun_out/r05/bench_2.err; echo "2 states rc=$?"
python benchmarks/api_overhead.py > gpurun_out/r05/api_overhead.txt 2>&1; echo "api rc=$?"
python bench.py --two-streams --no-cpu-baseline --no-extras > gpurun_out/r05/bench_two_streams.json 2>/dev/null; echo "2 streams rc=$?"
