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

torch is imported lazily: the single-GPU product path never needs it.
"""
import numpy as np

__all__ = ['shard_bounds', 'exchange_halos', 'allgather_matches', 'merge_rank_streams']


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

    def head_of(r):
        return edges[r][2:2 + int(edges[r][0])].astype(np.uint8)

    def tail_of(r):
        return edges[r][2 + halo:2 + halo + int(edges[r][1])].astype(np.uint8)
    # a shard shorter than the halo contributes all of itself (head == tail == shard): keep walking
    left_parts, need, r = [], halo, rank - 1
    while need > 0 and r >= 0:
        t = tail_of(r)[-need:] if need else tail_of(r)[:0]
        left_parts.insert(0, t)
        need -= len(t)
        r -= 1
    right_parts, need, r = [], halo, rank + 1
    while need > 0 and r < world:
        h = head_of(r)[:need]
        right_parts.append(h)
        need -= len(h)
        r += 1
    left = np.concatenate(left_parts) if left_parts else np.empty(0, np.uint8)
    right = np.concatenate(right_parts) if right_parts else np.empty(0, np.uint8)
    return left, right


_gather_cap = {}          # group -> agreed row capacity of the single-collective fast path


def allgather_matches(raw, group=None):
    """raw: this rank's stream, rows (start, end, dist, block) in global coordinates (list of tuples
    or structured/2-D numpy array).  -> (M_total, 4) int64 array in the reference's global order, on
    every rank.

    ONE collective per call in the common case: every rank contributes a fixed-capacity block
    [count, rows...]; the counts ride along, so no separate count exchange and a single device->host
    copy.  If some rank's count exceeds the agreed capacity every rank sees it in the gathered
    counts, the capacity is raised identically everywhere and the gather is repeated."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = _device_for(group)
    if isinstance(raw, np.ndarray) and raw.dtype.names:
        rows = np.stack([raw[f].astype(np.int64) for f in ('start', 'end', 'dist', 'block')], axis=1) \
            if len(raw) else np.empty((0, 4), np.int64)
    else:
        rows = np.asarray(raw, dtype=np.int64).reshape(-1, 4)
    key = id(group) if group is not None else 0
    while True:
        cap = _gather_cap.get(key, 4096)
        block = np.zeros((cap + 1, 4), dtype=np.int64)
        block[0, 0] = len(rows)
        fit = min(len(rows), cap)
        block[1:1 + fit] = rows[:fit]
        mine = torch.from_numpy(block).to(dev)
        gathered = torch.empty((world,) + tuple(mine.shape), dtype=torch.int64, device=dev)
        dist.all_gather(list(gathered.unbind(0)), mine, group=group)     # works on nccl and gloo
        host = gathered.cpu().numpy()
        counts = host[:, 0, 0]
        if int(counts.max()) <= cap:
            return merge_rank_streams([host[r, 1:1 + int(counts[r])] for r in range(world)])
        new_cap = cap
        while new_cap < int(counts.max()):
            new_cap *= 2
        _gather_cap[key] = new_cap
