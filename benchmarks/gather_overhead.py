"""Cost of tests/torch_glue.allgather_matches on one GPU (world = 1, nccl): staging + collective + merge."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from tests import torch_glue
n = 2409
raw = np.zeros(n, dtype=torch_glue.MATCH_DTYPE)
raw["start"] = np.sort(np.random.default_rng(1).integers(0, 1 << 30, n)); raw["end"] = raw["start"] + 20
raw["block"] = np.sort(np.random.default_rng(2).integers(0, 3, n))
for _ in range(20): torch_glue.allgather_matches(raw, as_array=True)
t0 = time.perf_counter(); N = 200
for _ in range(N): out = torch_glue.allgather_matches(raw, as_array=True)
print("allgather_matches: %.1f us/call (%d rows)" % ((time.perf_counter() - t0) / N * 1e6, len(out)))
st = fzd._gather_state[0]
def t(fn, name):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    torch.cuda.synchronize()
    print("  %-28s %.1f us" % (name, (time.perf_counter() - t0) / N * 1e6))
t(lambda: st["d_send"].copy_(st["h_send"], non_blocking=True), "H2D staging (async)")
t(lambda: dist.all_gather(list(st["d_recv"].unbind(0)), st["d_send"]), "all_gather (list api)")
t(lambda: dist.all_gather_into_tensor(st["d_recv"], st["d_send"]), "all_gather_into_tensor")
t(lambda: (st["h_recv"].copy_(st["d_recv"], non_blocking=True), torch.cuda.current_stream().synchronize()), "D2H + sync")
import ctypes
from fuzzysearch_amd import _native
lib = _native.load_library()
t(lambda: lib.fz_wire_pack(raw.__array_interface__["data"][0], n, st["cap"], st["send_ptr"]), "fz_wire_pack")
out = np.empty(st["cap"], dtype=torch_glue.MATCH_DTYPE); tot, top = ctypes.c_uint64(), ctypes.c_uint64()
t(lambda: lib.fz_wire_merge(st["recv_ptr"], 1, st["rows"], st["cap"], out.__array_interface__["data"][0], len(out),
                            ctypes.byref(tot), ctypes.byref(top)), "fz_wire_merge (1 rank)")
dist.destroy_process_group()
