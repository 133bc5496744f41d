"""Worker of tests/test_distributed_gloo.py: world_size ranks over gloo (CPU).  Each rank builds the
same global sequence, keeps only its shard + exchanged halos, searches the shard with the
host-compiled device logic (tests/host_emul.cpp: the code the GPU kernels run per candidate, with
everything outside the shard buffer poisoned), all-gathers the match lists and checks the merged
stream against the oracle run on the whole sequence."""
import ctypes
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from fuzzysearch_amd import distributed as fzd  # noqa: E402
from tests import torch_glue, workloads  # noqa: E402


class OutRec(ctypes.Structure):
    _fields_ = [("start", ctypes.c_int64), ("end", ctypes.c_int64), ("dist", ctypes.c_int32), ("block", ctypes.c_int32)]


def main():
    emul_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    L = ctypes.CDLL(emul_path)
    L.emul_search.restype = ctypes.c_int64
    L.emul_search.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64,
                              ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                              ctypes.POINTER(OutRec), ctypes.c_int64]
    ok = True
    for case, (n, m, k) in enumerate([(200000, 20, 2), (65537, 12, 1), (4099, 23, 5), (50, 9, 2), (7, 9, 2)]):
        seq = workloads.dna(n, 100 + case)
        pattern = workloads.dna(m, 200 + case)
        workloads.plant_variants(seq, pattern, 64, 300 + case)
        lo, hi = fzd.shard_bounds(n, world, rank)
        for b in range(1, world):                      # variants straddling every shard boundary
            cut = fzd.shard_bounds(n, world, b)[0]
            if m // 2 <= cut <= n - m:
                seq[cut - m // 2:cut - m // 2 + m] = pattern
        if case == 0:                                  # ... and the bench's plants at deltas {-m-k ... +1} (equal shards)
            plants = workloads.boundary_plants(m, k, n // world, world)
            workloads.apply_plants(seq, 0, plants, pattern)
        p, halo = pattern.tobytes(), m + k
        shard = seq[lo:hi].copy()                      # all this rank keeps of the sequence
        left, right = torch_glue.exchange_halos(shard, halo)
        assert bytes(left) == seq[max(0, lo - halo):lo].tobytes() and bytes(right) == seq[hi:hi + halo].tobytes()
        buf = np.concatenate([left, shard, right])
        buf_off = lo - len(left)
        # the emulator wants the global array for addressing; give it one that is poison outside buf
        fake = np.full(n, 0xEE, dtype=np.uint8)
        fake[buf_off:buf_off + len(buf)] = buf
        cap = 1 << 16
        out = (OutRec * cap)()
        c = L.emul_search(1, p, m, fake.tobytes(), n, k, buf_off, len(buf), lo, hi, out, cap)
        assert 0 <= c <= cap
        mine = [(out[i].start, out[i].end, out[i].dist, out[i].block) for i in range(c)]
        merged = torch_glue.allgather_matches(mine)
        exp = oracle.lev_ngrams_raw(p, seq.tobytes(), k)
        if case == 0 and n % world == 0:
            found = {(int(a), int(b), int(c)) for (a, b, c, _g) in merged}
            assert all((q, q + m, 0) in found for q in plants), "boundary plants missing"
        if [tuple(int(x) for x in r) for r in merged] != exp:
            ok = False
            print("rank %d case %d MISMATCH: %d vs %d" % (rank, case, len(merged), len(exp)), flush=True)
        elif rank == 0:
            print("case %d ok: %d raw matches, %d local on rank 0" % (case, len(exp), len(mine)), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("RANK %d %s" % (rank, "PASS" if ok else "FAIL"), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
