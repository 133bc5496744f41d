cd ${GRAFT_REPO_ROOT:-.}
python -c "from tests import mock_rccl; mock_rccl.build()"
export FZ_RCCL_LIB=$PWD/tests/libmock_rccl.so FZ_RENDEZVOUS_KEY=lf$$ WORLD_SIZE=3 MASTER_ADDR=127.0.0.1 MASTER_PORT=29999
export FZMOCK_FAIL_ALLGATHER=30 FZMOCK_FAIL_RANK=1 FZ_COMM_TIMEOUT_MS=400 FZ_MOCK_RCCL_TIMEOUT_S=3
for r in 1 2; do RANK=$r LOCAL_RANK=$r python bench.py --gpus 3 --mib 128 --steps 12 --warmup 3 --settle-ms 300 --no-cpu-baseline > gpurun_out/lf_rank$r.out 2> gpurun_out/lf_rank$r.err & done
RANK=0 LOCAL_RANK=0 python bench.py --gpus 3 --mib 128 --steps 12 --warmup 3 --settle-ms 300 --no-cpu-baseline > gpurun_out/lf_rank0.out 2> gpurun_out/lf_rank0.err
wait
cat gpurun_out/lf_rank0.out; echo; tail -5 gpurun_out/lf_rank0.err; echo ---; tail -8 gpurun_out/lf_rank1.err; ls /dev/shm | head
