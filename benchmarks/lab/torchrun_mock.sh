cd ${GRAFT_REPO_ROOT:-.}
python -c "from tests import mock_rccl; mock_rccl.build()"
export FZ_RCCL_LIB=$PWD/tests/libmock_rccl.so
FZMOCK_FAIL_ALLGATHER=40 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 4 --steps 20 --warmup 5 --mib 256 --no-cpu-baseline > gpurun_out/tr_mid.out 2> gpurun_out/tr_mid.err; echo "mid-run failure rc=$?"
FZMOCK_HANG_INIT_S=100 FZ_COMM_INIT_TIMEOUT_S=3 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29614 bench.py --gpus 4 --steps 20 --warmup 5 --mib 256 --no-cpu-baseline > gpurun_out/tr_hang.out 2> gpurun_out/tr_hang.err; echo "hanging init rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/tr_mid.out", "gpurun_out/tr_hang.out"):
    n = 0
    for l in open(f):
        if l.startswith("{"):
            n += 1
            d = json.loads(l); print(f, d["n_gpus"], d["rccl_ranks"], d["value"], d.get("collective_error")[:150], d.get("boundary_plants_found"), d["stream_in_reference_order"])
    print(f, "lines:", n)
PY
ls /dev/shm | grep -c "fz_fb" 
