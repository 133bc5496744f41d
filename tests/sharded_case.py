"""Construction of the sharded configs[4] test case (tests/test_gpu_multi_device.py) and of its COMPLETE expected
stream from cheap oracle runs: the global sequence is copies of one block, every place that differs is saved as a
small window while the shards are built.  tests/test_sharded_case.py checks the construction itself on the CPU
(against the oracle run on the whole of a small sequence)."""
import numpy as np

import oracle
from tests import workloads

WIN = 4096          # bytes saved on both sides of every special place
MARGIN = 64         # > m + 2k: a match that starts this far inside a window only depends on bytes of the window


def build(world, shard_bytes, pattern, k, region_bytes, BLOCK):
    """-> (fill, windows, edge): fill(r, out) for iter_shard_buffers; `windows` collects (lo, hi, bytes) of every
    special place as the shards go by (a window across a shard boundary is completed by the next shard)."""
    m = len(pattern)
    base = workloads.dna(BLOCK, 900)
    edge = workloads.boundary_plants(m, k, shard_bytes, world)
    tiles = shard_bytes // BLOCK
    n = world * shard_bytes
    windows = {}                                             # lo -> [lo, hi, bytearray]
    spans = []
    for seam in range(0, n + 1, BLOCK):                      # tile seams, shard boundaries, both ends
        spans.append((max(0, seam - WIN), min(n, seam + WIN)))
    regions = []
    for r in range(world):                                   # one freshly generated region per shard, inside tile 5
        lo = r * shard_bytes + min(tiles - 1, 5) * BLOCK + (BLOCK >> 6) * (r + 1)
        regions.append((lo, lo + region_bytes))
        spans.append((lo - WIN, lo + region_bytes + WIN))

    def fill(r, out):
        out.reshape(tiles, BLOCK)[:] = base
        workloads.apply_plants(out, r * shard_bytes, edge, pattern)
        lo, hi = regions[r]
        reg = workloads.dna(region_bytes, 1000 + r)
        workloads.plant_variants(reg, pattern, 64, 40 + r)
        out[lo - r * shard_bytes:hi - r * shard_bytes] = reg
        g0 = r * shard_bytes
        for (a, b) in spans:                                 # save the part of every window that lies in this shard
            x, y = max(a, g0), min(b, g0 + shard_bytes)
            if x < y:
                w = windows.setdefault(a, [a, b, bytearray(b - a)])
                w[2][x - a:y - a] = out[x - g0:y - g0].tobytes()
    return base, fill, windows, edge


def expected(base, windows, n, p, k):
    BLOCK = len(base)
    """The complete expected raw stream as a sorted list of (start, end, dist, block)."""
    inner = []                                               # [lo, hi) of starts decided by a window
    exp = []
    for (a, b, data) in sorted(windows.values()):
        lo = a if a == 0 else a + MARGIN
        hi = b if b == n else b - MARGIN
        inner.append((lo, hi))
        for (s, e, d, g) in oracle.lev_ngrams_raw(p, bytes(data), k):
            if lo <= s + a < hi:
                exp.append((s + a, e + a, d, g))
    inner.sort()
    assert all(inner[i][1] <= inner[i + 1][0] for i in range(len(inner) - 1)), 'special windows overlap'
    los = np.array([x for x, _ in inner]); his = np.array([y for _, y in inner])
    exp0 = [r for r in oracle.lev_ngrams_raw(p, base.tobytes(), k) if MARGIN <= r[0] and r[1] <= BLOCK - MARGIN]
    s0 = np.array([r[0] for r in exp0], dtype=np.int64)
    for c in range(n // BLOCK):
        starts = s0 + c * BLOCK
        j = np.searchsorted(los, starts, side='right') - 1   # the window that could hold this start
        covered = (j >= 0) & (starts < his[np.maximum(j, 0)])
        for i in np.flatnonzero(~covered):
            s, e, d, g = exp0[i]
            exp.append((s + c * BLOCK, e + c * BLOCK, d, g))
    return sorted(exp)
