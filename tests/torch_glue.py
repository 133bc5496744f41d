"""torch.distributed glue around the C-ABI's wire format — TEST / BENCH infrastructure, not product code (moved out of
fuzzysearch_amd/distributed.py in round 5).  The product's exchange step is RCCL behind the C-ABI (fz_comm_*); this module
moves the same data through torch collectives so that (a) the N > 1 data path — shards, halos, ownership, merge order —
runs on CPUs with the "gloo" backend (tests/test_distributed_gloo.py, world 2 and 3) and (b) `FZ_BENCH_TORCH=1 bench.py`
can be launched under torch.distributed.run with the "nccl" (= RCCL) backend.  torch is imported lazily."""
import numpy as np

from fuzzysearch_amd.distributed import _halos_from_edges


def _device_for(group):
    import torch
    import torch.distributed as dist
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' \
        else torch.device('cpu')


def exchange_halos(shard, halo, group=None):
    """All ranks hold equal-length... or ragged shards of one global sequence in rank order.  Returns
    (left, right): the last `halo` bytes of the previous rank's shard and the first `halo` bytes of
    the next rank's (empty at the ends).  One small all_gather at load time."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _device_for(group)
    shard = np.asarray(shard, dtype=np.uint8)
    edge = np.zeros(2 * halo + 2, dtype=np.int64)          # [n_head, n_tail, head bytes..., tail bytes...]
    head, tail = shard[:halo], shard[-halo:] if halo else shard[:0]
    edge[0], edge[1] = len(head), len(tail)
    edge[2:2 + len(head)] = head
    edge[2 + halo:2 + halo + len(tail)] = tail
    mine = torch.from_numpy(edge).to(dev)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    edges = [g.cpu().numpy() for g in gathered]

    return _halos_from_edges([(e[2:2 + int(e[0])].astype(np.uint8), e[2 + halo:2 + halo + int(e[1])].astype(np.uint8))
                              for e in edges], rank, halo)


MATCH_DTYPE = np.dtype([("start", "<i8"), ("end", "<i8"), ("dist", "<i4"), ("block", "<i4")])   # = fz_match

_gather_state = {}        # group -> dict(cap, host staging tensors, device tensors)


def _as_match_array(raw):
    if isinstance(raw, np.ndarray) and raw.dtype == MATCH_DTYPE:
        return np.ascontiguousarray(raw)
    rows = np.asarray(raw, dtype=np.int64).reshape(-1, 4)
    out = np.empty(len(rows), dtype=MATCH_DTYPE)
    out["start"], out["end"], out["dist"], out["block"] = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]
    return out


def merge_rank_arrays(parts, block_counts=None):
    """parts[r] = rank r's fz_match array in reference order (block-major, index ascending within a
    block), ranks owning ascending index ranges -> one array in the global reference order: for every
    block, the ranks' segments of that block back to back (fz_merge_ranks: no sort, O(total) copies).
    block_counts[r][g] (optional) = number of rank r's records of block g."""
    world = len(parts)
    parts = [np.ascontiguousarray(p, dtype=MATCH_DTYPE) for p in parts]
    if block_counts is None:
        nb = max([int(p["block"][-1]) + 1 for p in parts if len(p)] or [0])
        block_counts = np.zeros((world, nb), dtype=np.uint64)
        for r, p in enumerate(parts):
            if len(p):
                block_counts[r, :] = np.bincount(p["block"], minlength=nb)[:nb]
    addrs = [p.__array_interface__["data"][0] for p in parts]
    return _merge_native(addrs, [len(p) for p in parts], block_counts)


def _merge_native(addrs, counts, block_counts):
    """addrs[r] = address of rank r's first fz_match record (24-byte rows), counts[r] = how many."""
    import ctypes
    from fuzzysearch_amd import _native
    world = len(addrs)
    bc = np.ascontiguousarray(block_counts, dtype=np.uint64).reshape(world, -1)
    cnt = np.asarray(counts, dtype=np.uint64)
    out = np.empty(int(cnt.sum()), dtype=MATCH_DTYPE)
    ptrs = (ctypes.c_void_p * world)(*addrs)
    _native._check(_native.load_library().fz_merge_ranks(
        ptrs, cnt.__array_interface__["data"][0], bc.__array_interface__["data"][0], world, bc.shape[1],
        out.__array_interface__["data"][0]))
    return out


WIRE_HEADER_ROWS = 65     # FZ_WIRE_HEADER_ROWS of include/fzhip.h: count, nblocks, 256 per-block counts


def allgather_matches(raw, group=None, as_array=False):
    """raw: this rank's stream in global coordinates — the fz_match structured array of
    Engine.lev_ngrams(..., as_array=True) or a list of (start, end, dist, block) tuples.
    -> the merged stream in the reference's global order on every rank, as an (M, 4) int64 array
    (or the fz_match structured array with as_array=True).

    ONE collective per call in the common case: every rank contributes a fixed-capacity block in the
    16-byte wire format of include/fzhip.h (header: count + per-block counts, then the records; the
    counts ride along, so there is no separate count exchange and the merge needs no sort), packed and
    merged by the C library, staged through persistent pinned host buffers with a single stream
    synchronisation.  If some rank's count exceeds the agreed capacity every rank sees it in the
    gathered headers, the capacity is raised identically everywhere and the gather is repeated; the
    capacity also follows the counts down."""
    import ctypes
    import torch
    import torch.distributed as dist
    from fuzzysearch_amd import _native
    lib = _native.load_library()
    world = dist.get_world_size(group)
    dev = _device_for(group)
    mine = _as_match_array(raw)
    key = group if group is not None else 0      # (the dict holds the group itself: its identity cannot be recycled)
    while True:
        st = _gather_state.get(key)
        # the state holds the group object (so its id cannot be handed to another group while the entry exists)
        # and is rebuilt if world size or device differ from what it was built for
        if st is not None and (st.get("group") is not group or st.get("dev", dev) != dev):
            st = None
        if st is None or st["world"] != world:
            cap = st["cap"] if st else 4096
            pin = dev.type == "cuda"
            rows = WIRE_HEADER_ROWS + cap
            st = {"cap": cap, "world": world, "rows": rows, "group": group, "dev": dev,
                  "h_send": torch.zeros((rows, 2), dtype=torch.int64, pin_memory=pin),
                  "h_recv": torch.zeros((world, rows, 2), dtype=torch.int64, pin_memory=pin)}
            st["send_ptr"], st["recv_ptr"] = st["h_send"].data_ptr(), st["h_recv"].data_ptr()
            if pin:
                st["d_send"] = torch.zeros((rows, 2), dtype=torch.int64, device=dev)
                st["d_recv"] = torch.zeros((world, rows, 2), dtype=torch.int64, device=dev)
            _gather_state[key] = st
        cap = st["cap"]
        _native._check(lib.fz_wire_pack(mine.__array_interface__["data"][0], len(mine), cap, st["send_ptr"]))
        if dev.type == "cuda":
            st["d_send"].copy_(st["h_send"], non_blocking=True)
            dist.all_gather_into_tensor(st["d_recv"], st["d_send"], group=group)
            st["h_recv"].copy_(st["d_recv"], non_blocking=True)
            torch.cuda.current_stream().synchronize()
        else:
            dist.all_gather(list(st["h_recv"].unbind(0)), st["h_send"], group=group)
        total, top = ctypes.c_uint64(0), ctypes.c_uint64(0)
        # first pass: totals only (out_cap 0 is fine when a re-gather is needed or nothing matched)
        merged = np.empty(world * cap, dtype=MATCH_DTYPE)
        _native._check(lib.fz_wire_merge(st["recv_ptr"], world, st["rows"], cap, merged.__array_interface__["data"][0],
                                         len(merged), ctypes.byref(total), ctypes.byref(top)))
        if top.value <= cap:
            merged = merged[:total.value]
            # keep the exchanged block near the size that is used (every rank sees the same counts, so
            # every rank resizes identically): the collective and the D2H copy move `cap` rows per rank
            want = max(256, -(-(top.value + top.value // 8) // 128) * 128)
            if want * 4 <= cap * 3:
                _gather_state[key] = {"cap": want, "world": -1, "group": group, "dev": dev}
            if as_array:
                return merged
            out = np.empty((len(merged), 4), dtype=np.int64)
            out[:, 0], out[:, 1], out[:, 2], out[:, 3] = merged["start"], merged["end"], merged["dist"], merged["block"]
            return out
        new_cap = cap
        while new_cap < top.value:
            new_cap *= 2
        _gather_state[key] = {"cap": new_cap, "world": -1, "group": group, "dev": dev}   # rebuild buffers at the new capacity
