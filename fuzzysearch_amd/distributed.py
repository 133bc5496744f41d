"""Sharded search across the GPUs of a node, one process per GPU (torch.distributed; the "nccl"
backend is RCCL over xGMI on ROCm, "gloo" works on CPUs for tests).

The path shards with ONE exchange step: every raw match is a pure function of its
n-gram hit index and the bytes within (m + k) of it, so
  * the global sequence [0, N) is cut into contiguous shards, rank r owning hits with
    own_lo <= idx < own_hi;
  * rank r holds its shard plus (m + k) halo bytes of each neighbour (exchanged once, at load);
  * every rank scans its shard with no data-path collective (fz_seq_upload_shard + fz_lev_ngrams,
    results in GLOBAL coordinates);
  * the per-rank raw match lists are all-gathered (counts, then records padded to the max count —
    KB-scale, latency-bound; xGMI bandwidth is irrelevant) and concatenated in rank order; a stable
    sort on the block index restores the reference's emission order because ranks own ascending
    index ranges.
Consolidation then runs once on the gathered list (overlap groups can span shard boundaries).

Two ways to run it:
  * torch-free (the product path): the collective lives behind the C-ABI (fz_comm_*, RCCL linked into libfzhip.so).
    `init_engine_from_env()` turns the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK, as set by
    torch.distributed.run, mpirun wrappers, ...) into a single-device engine that has joined the job's communicator:
    rank 0 creates the RCCL unique id and hands it over through a small rendezvous file; from then on
    engine.lev_ngrams() is collective and returns the merged global stream.  One process with several GPUs needs no
    rendezvous at all: Engine([0, 1, ...]).comm_init_all().
  * torch.distributed (optional launcher glue, and the CPU test double: "gloo"): exchange_halos / allgather_matches
    below move the same data through torch collectives; torch is imported lazily and never by the product path.
"""
import os
import time

import numpy as np

__all__ = ['shard_bounds', 'exchange_halos', 'allgather_matches', 'merge_rank_streams',
           'init_engine_from_env', 'local_device', 'share_blob', 'exchange_halos_native']

_rdzv_seq = [0]


def _job_key():
    """What every rank of one job on this node agrees on and no other live job shares: the launcher process (all
    ranks are its children) with its start time, the rendezvous port and the restart count.  FZ_RENDEZVOUS_KEY
    overrides (ranks started by different parents)."""
    key = os.environ.get("FZ_RENDEZVOUS_KEY")
    if key:
        return key
    ppid = os.getppid()
    try:
        with open("/proc/%d/stat" % ppid) as f:
            start = f.read().rsplit(")", 1)[1].split()[19]          # field 22: starttime
    except Exception:
        start = "0"
    return "%d_%s_%s_%s" % (ppid, start, os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))


def share_blob(make_blob, world, rank, timeout=300.0, directory=None):
    """Rank 0 calls make_blob() and every rank returns those bytes: a file written atomically under a name that only
    the ranks of this job derive (single node, as the bench contract has it).  Torch-free rendezvous for the RCCL
    unique id; every call of a process uses a fresh name, rank 0 removes the file once every rank has confirmed."""
    seq = _rdzv_seq[0]
    _rdzv_seq[0] += 1
    if world == 1:
        return make_blob()
    directory = directory or os.environ.get("FZ_RENDEZVOUS_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    base = os.path.join(directory, "fz_rdzv_%s_%d" % (_job_key(), seq))
    deadline = time.monotonic() + timeout
    if rank == 0:
        blob = make_blob()
        tmp = base + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(blob)
        os.replace(tmp, base)
        acks = [base + ".ack%d" % r for r in range(1, world)]
        while not all(os.path.exists(a) for a in acks):
            if time.monotonic() > deadline:
                raise TimeoutError("rendezvous: ranks %r never read %s" % ([r for r in range(1, world) if not os.path.exists(base + ".ack%d" % r)], base))
            time.sleep(0.002)
        for a in acks + [base]:
            try:
                os.remove(a)
            except OSError:
                pass
        return blob
    while True:
        try:
            with open(base, "rb") as f:
                blob = f.read()
            break
        except FileNotFoundError:
            if time.monotonic() > deadline:
                raise TimeoutError("rendezvous: rank 0 never wrote %s" % base)
            time.sleep(0.002)
    with open(base + ".ack%d" % rank, "wb"):
        pass
    return blob


def local_device(local_rank):
    """The HIP device of this rank: LOCAL_RANK, unless the launcher already narrowed the visible devices per rank
    (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES: then fewer devices are visible than there are local ranks)."""
    import ctypes
    from . import _native
    n = ctypes.c_int(0)
    _native._check(_native.load_library().fz_device_count(ctypes.byref(n)))
    return local_rank if local_rank < n.value else local_rank % max(1, n.value)


def init_engine_from_env(engine=None):
    """One process per GPU under any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run does):
    -> (engine, world, rank), the engine on device LOCAL_RANK and a member of the job's RCCL communicator.  No torch."""
    from . import _native
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if engine is None:
        engine = _native.Engine([local_device(local_rank)])
    uid = share_blob(engine.comm_unique_id, world, rank)
    engine.comm_init_rank(uid, world, rank)
    return engine, world, rank


def exchange_halos_native(engine, shard, halo):
    """exchange_halos() over the engine's own communicator (fz_comm_allgather): -> (left, right)."""
    world, rank, _ = engine.comm_info()
    shard = np.asarray(shard, dtype=np.uint8)
    head, tail = shard[:halo], (shard[-halo:] if halo else shard[:0])
    blob = np.zeros(16 + 2 * halo, dtype=np.uint8)
    blob[:16].view(np.uint64)[:] = (len(head), len(tail))
    blob[16:16 + len(head)] = head
    blob[16 + halo:16 + halo + len(tail)] = tail
    parts = [np.frombuffer(b, dtype=np.uint8) for b in engine.comm_allgather(blob.tobytes())]
    return _halos_from_edges([(p[16:16 + int(p[:8].view(np.uint64)[0])], p[16 + halo:16 + halo + int(p[8:16].view(np.uint64)[0])])
                              for p in parts], rank, halo)


def _halos_from_edges(edges, rank, halo):
    """edges[r] = (first, last) `halo` bytes of rank r's shard (all of it when it is shorter) -> (left, right) of
    `rank`: a shard shorter than the halo contributes all of itself, so keep walking."""
    world = len(edges)
    left_parts, need, r = [], halo, rank - 1
    while need > 0 and r >= 0:
        t = edges[r][1][-need:] if need else edges[r][1][:0]
        left_parts.insert(0, t)
        need -= len(t)
        r -= 1
    right_parts, need, r = [], halo, rank + 1
    while need > 0 and r < world:
        h = edges[r][0][:need]
        right_parts.append(h)
        need -= len(h)
        r += 1
    left = np.concatenate(left_parts) if left_parts else np.empty(0, np.uint8)
    right = np.concatenate(right_parts) if right_parts else np.empty(0, np.uint8)
    return left.astype(np.uint8), right.astype(np.uint8)


def shard_bounds(n, world, rank):
    """Contiguous, near-equal ownership ranges: -> (own_lo, own_hi)."""
    base, rem = divmod(n, world)
    lo = base * rank + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def merge_rank_streams(streams):
    """Concatenate per-rank raw streams ((start, end, dist, block) rows, each already in reference
    order, ranks owning ascending index ranges) into the global reference order."""
    arrs = [np.asarray(s, dtype=np.int64).reshape(-1, 4) for s in streams]
    allm = np.concatenate(arrs) if arrs else np.empty((0, 4), np.int64)
    if len(allm) == 0:
        return allm
    return allm[np.argsort(allm[:, 3], kind='stable')]


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
    from . import _native
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
    from . import _native
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
