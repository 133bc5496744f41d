cd ${GRAFT_REPO_ROOT:-.}
# two ranks on ONE GPU with the REAL librccl: RCCL refuses (duplicate GPU) or never completes — what does bench.py print?
FZ_COMM_INIT_TIMEOUT_S=60 FZ_COMM_TIMEOUT_MS=5000 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 20 --warmup 5 --mib 256 --no-cpu-baseline > gpurun_out/tr_real.out 2> gpurun_out/tr_real.err; echo "real rccl, 2 ranks on 1 GPU: rc=$?"
python - <<'PY'
import json
n = 0
for l in open("gpurun_out/tr_real.out"):
    if l.startswith("{"):
        n += 1
        d = json.loads(l); print(d["n_gpus"], d["rccl_ranks"], d["value"], str(d.get("collective_error"))[:400], d.get("boundary_plants_found"), d["stream_in_reference_order"])
print("lines:", n)
PY
grep -i "nccl\|rccl\|error" gpurun_out/tr_real.err | head -12
