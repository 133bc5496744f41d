"""CPU check of the sharded test construction (tests/sharded_case.py): the expected stream assembled from the
block's matches plus the saved windows equals the oracle run on the WHOLE (small) sequence, and the shard
buffers handed to fz_seq_add_shard are the right slices of it."""
import numpy as np

import oracle
from tests import sharded_case, workloads


def test_expected_stream_equals_the_oracle_on_the_whole_sequence():
    block, world, tiles = 1 << 20, 3, 2
    shard_bytes = tiles * block
    pattern = workloads.dna(20, 1)
    p, m, k = pattern.tobytes(), 20, 2
    n = world * shard_bytes
    base, fill, windows, edge = sharded_case.build(world, shard_bytes, pattern, k, 1 << 16, block)
    full = np.empty(n, dtype=np.uint8)
    for r, buf, off, lo, hi in workloads.iter_shard_buffers(world, shard_bytes, m + k, fill):
        full[lo:hi] = buf[lo - off:hi - off]
        assert off == max(0, lo - (m + k)) and off + len(buf) == min(n, hi + m + k)
    # (halos agree with the neighbours' own bytes)
    for r, buf, off, lo, hi in workloads.iter_shard_buffers(world, shard_bytes, m + k, sharded_case.build(world, shard_bytes, pattern, k, 1 << 16, block)[1]):
        assert np.array_equal(buf, full[off:off + len(buf)])
    exp = sharded_case.expected(base, windows, n, p, k)
    whole = sorted(oracle.lev_ngrams_raw(p, full.tobytes(), k))
    assert exp == whole
    found = {(s, e, d) for (s, e, d, _g) in whole}
    assert len(edge) == 5 and all((q, q + m, 0) in found for q in edge)
