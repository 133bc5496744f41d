"""Sharded search across the GPUs of a node, one process per GPU (RCCL over xGMI behind the C-ABI; any launcher that
sets RANK / WORLD_SIZE / LOCAL_RANK, e.g. torch.distributed.run — torch itself is not imported).

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
  * torch-free (the product path): the collective lives behind the C-ABI (fz_comm_*, RCCL loaded by libfzhip.so on first use).
    `init_engine_from_env()` turns the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK, as set by
    torch.distributed.run, mpirun wrappers, ...) into a single-device engine that has joined the job's communicator:
    rank 0 creates the RCCL unique id and hands it over through a small rendezvous file; from then on
    engine.lev_ngrams() is collective and returns the merged global stream.  One process with several GPUs needs no
    rendezvous at all: Engine([0, 1, ...]).comm_init_all().
  * torch.distributed is NOT part of this package: the CPU test double of the exchange step (gloo, world 2 / 3) and the
    optional FZ_BENCH_TORCH=1 launcher glue live in tests/torch_glue.py (exchange_halos / allgather_matches over torch
    collectives, the same wire format and merge: fz_wire_pack / fz_wire_merge of the C-ABI).
"""
import os
import time

import numpy as np

__all__ = ['shard_bounds', 'merge_rank_streams', 'init_engine_from_env', 'local_device', 'share_blob',
           'exchange_halos_native', 'halos_from_edges', 'FileRendezvous']

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


class FileRendezvous(object):
    """All-gather and barrier of small blobs between the ranks of ONE node through files (/dev/shm), for the moments a job
    has no working collective library: `bench.py` under a launcher falls back to it when RCCL cannot set the communicator up,
    fails in the middle of a run or never completes (every rank searches its shard, the streams are merged once through
    here and the line says so).  Milliseconds per exchange — a way to agree and to hand results over, not a data path.
    Every rank must make the same sequence of calls; every wait has a deadline (TimeoutError)."""

    def __init__(self, world, rank, tag="fb", timeout=300.0, directory=None):
        self.world, self.rank, self.timeout = int(world), int(rank), float(timeout)
        directory = directory or os.environ.get("FZ_RENDEZVOUS_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
        self.dir = os.path.join(directory, "fz_%s_%s" % (tag, _job_key()))
        os.makedirs(self.dir, exist_ok=True)
        self.step = 0
        self._mine = []

    def _name(self, step, rank):
        return os.path.join(self.dir, "%d_%d" % (step, rank))

    def allgather(self, blob):
        """-> [rank 0's blob, rank 1's, ...] on every rank."""
        step = self.step
        self.step += 1
        mine = self._name(step, self.rank)
        tmp = mine + ".tmp"
        with open(tmp, "wb") as f:
            f.write(bytes(blob))
        os.replace(tmp, mine)
        self._mine.append(mine)
        deadline = time.monotonic() + self.timeout
        out = []
        for r in range(self.world):
            name = self._name(step, r)
            while True:
                try:
                    with open(name, "rb") as f:
                        out.append(f.read())
                    break
                except FileNotFoundError:
                    if time.monotonic() > deadline:
                        raise TimeoutError("file rendezvous %s: rank %d never arrived at step %d" % (self.dir, r, step))
                    time.sleep(0.0005)
        # whoever is past step n has seen every rank's file of step n, which a rank writes after it has read all of step n - 1:
        # this rank's file of step n - 1 has been read by everybody
        while len(self._mine) > 1:
            try:
                os.remove(self._mine.pop(0))
            except OSError:
                pass
        return out

    def barrier(self):
        self.allgather(b"")

    def close(self):
        """Collective, once, last: every rank leaves a `done` mark behind its last read; rank 0 waits for all of them and
        removes the directory (a rank cannot know when ITS last file has been read by everybody — rank 0 can)."""
        done = os.path.join(self.dir, "done_%d" % self.rank)
        with open(done, "wb"):
            pass
        self._mine = []
        if self.rank != 0:
            return
        deadline = time.monotonic() + self.timeout
        while not all(os.path.exists(os.path.join(self.dir, "done_%d" % r)) for r in range(self.world)):
            if time.monotonic() > deadline:
                return
            time.sleep(0.001)
        for name in os.listdir(self.dir):
            try:
                os.remove(os.path.join(self.dir, name))
            except OSError:
                pass
        try:
            os.rmdir(self.dir)
        except OSError:
            pass


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
    return halos_from_edges([(p[16:16 + int(p[:8].view(np.uint64)[0])], p[16 + halo:16 + halo + int(p[8:16].view(np.uint64)[0])])
                              for p in parts], rank, halo)


def halos_from_edges(edges, rank, halo):
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


_halos_from_edges = halos_from_edges
