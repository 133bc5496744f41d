mkdir -p gpurun_out/r06
python bench.py > gpurun_out/r06/bench_n1.json 2> gpurun_out/r06/bench_n1.err; echo "n1 rc=$?"
python -c "from tests import mock_rccl; mock_rccl.build()"
FZ_RCCL_LIB=$PWD/tests/libmock_rccl.so FZ_DEVICES=0,0,0,0,0,0,0,0 python bench.py --gpus 8 --mib 1024 --no-cpu-baseline > gpurun_out/r06/bench_8_ranks_stand_in.json 2> gpurun_out/r06/bench_8.err; echo "8 ranks rc=$?"
FZMOCK_STALL_ALLGATHER=60:5000 FZ_COMM_TIMEOUT_MS=500 FZ_RCCL_LIB=$PWD/tests/libmock_rccl.so FZ_DEVICES=0,0,0,0,0,0,0,0 python bench.py --gpus 8 --mib 256 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06/bench_8_ranks_stalled_collective.json 2> gpurun_out/r06/bench_8_stall.err; echo "8 ranks, stalled collective rc=$?"
FZ_DEVICES=0,0 python bench.py --gpus 2 --no-cpu-baseline > gpurun_out/r06/bench_two_device_states.json 2> gpurun_out/r06/bench_2.err; echo "2 states rc=$?"
python benchmarks/api_overhead.py > gpurun_out/r06/api_overhead.txt 2>&1; echo "api rc=$?"
python benchmarks/subs_dense.py > gpurun_out/r06/subs_dense_new.txt 2>&1; FZ_NO_BITS=1 python benchmarks/subs_dense.py > gpurun_out/r06/subs_dense_old.txt 2>&1; echo "subs dense rc=$?"
