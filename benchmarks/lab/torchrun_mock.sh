cd ${GRAFT_REPO_ROOT:-.}
python -c "from tests import mock_rccl; mock_rccl.build()"
export FZ_RCCL_LIB=$PWD/tests/libmock_rccl.so
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --mib 256 --no-cpu-baseline > gpurun_out/tr_ok.out 2> gpurun_out/tr_ok.err; echo "ok rc=$?"
FZMOCK_FAIL_INIT=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 20 --warmup 5 --mib 256 --no-cpu-baseline > gpurun_out/tr_fail.out 2> gpurun_out/tr_fail.err; echo "fail-init rc=$?"
grep -c "^{" gpurun_out/tr_ok.out gpurun_out/tr_fail.out
python - <<'PY'
import json
for f in ("gpurun_out/tr_ok.out", "gpurun_out/tr_fail.out"):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f, d["n_gpus"], d["rccl_ranks"], d["value"], d.get("collective_error"), d["boundary_plants_found"], d["stream_in_reference_order"])
PY
ls /dev/shm | grep -c fz_ ; tail -3 gpurun_out/tr_fail.err
