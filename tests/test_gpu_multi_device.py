"""-m gpu: BASELINE configs[4] at its stated size — 32 GiB of DNA in eight 4 GiB shards — as eight device states of
ONE torch-free context on one MI355X (8 x 4 GiB fits its HBM), through the same code path `bench.py --gpus 8` takes
(fz_seq_new / fz_seq_add_shard, shard by shard: never a 32 GiB host array).  Shards 1..7 start beyond 2^32.

Size-independent property that keeps the oracle cheap: the global sequence is 64 copies of one 512 MiB block, so
away from what was overwritten the matches inside copy c are those of copy 0 shifted by c * 512 MiB; every place
that differs (tile seams, shard boundaries with their planted copies, one freshly generated 1 MiB region per shard,
both ends of the sequence) is saved as a small window while the shards are built and the oracle runs on the window.
The union of the two is the COMPLETE expected stream: the merged GPU stream must equal it as a multiset (no match
lost at a boundary, none produced twice by two shards)."""
import numpy as np
import pytest

import oracle
from tests import sharded_case, workloads

pytestmark = pytest.mark.gpu

GIB = 1 << 30
BLOCK = 512 << 20


def _run(engine_cls, world, shard_mib, region_bytes=1 << 20):
    shard_bytes = shard_mib << 20
    pattern = workloads.dna(20, 1)
    p, m, k = pattern.tobytes(), 20, 2
    n = world * shard_bytes
    base, fill, windows, edge = sharded_case.build(world, shard_bytes, pattern, k, region_bytes, BLOCK)
    eng = engine_cls([0] * world)
    try:
        h = eng.new_sequence(n)
        for r, buf, off, lo, hi in workloads.iter_shard_buffers(world, shard_bytes, m + k, fill):
            eng.add_shard(h, r, buf, off, lo, hi)
            if r:
                assert off + (m + k) == lo and (r == 0 or off > 0)
        got_arr = eng.lev_ngrams(h, p, k, as_array=True)
        st = eng.stats()
        ms = eng.device_ms()
        # the two-deep pipeline over all shards delivers the same stream
        eng.lev_ngrams_begin(h, p, k)
        eng.lev_ngrams_begin(h, p, k)
        again1 = eng.lev_ngrams_end(as_array=True)
        again2 = eng.lev_ngrams_end(as_array=True)
        exact = eng.search_exact(h, p)
        h.release()
    finally:
        eng.close()
    assert np.array_equal(again1, got_arr) and np.array_equal(again2, got_arr)
    got = [tuple(int(x) for x in r) for r in got_arr.tolist()]
    assert st["bytes_scanned"] >= n and st["n_devices"] == world and len(ms) == world and min(ms) > 0
    blocks = [g for (_s, _e, _d, g) in got]
    assert blocks == sorted(blocks)                          # block-major: the reference's emission order
    exp = sharded_case.expected(base, windows, n, p, k)
    assert sorted(got) == exp
    found = {(s, e, d) for (s, e, d, _g) in got}
    assert all((q, q + m, 0) in found for q in edge), "a copy planted at a shard boundary is missing"
    assert set(edge) <= set(exact)
    assert exact == sorted(set(exact))
    if shard_bytes > (1 << 32) // 2:
        assert any(s > (1 << 32) for (s, _e, _d, _g) in got)
    return len(got), len(edge)


def test_multi_device_small_shards():
    """The same construction at 3 x 1 GiB... of 512 MiB tiles is too slow for a smoke: 3 shards of 1 tile each."""
    from fuzzysearch_amd import _native
    n_got, n_edge = _run(_native.Engine, 3, 512, region_bytes=1 << 18)
    assert n_got > 400 and n_edge == 5


def test_config4_32gib_eight_shards_on_one_gpu():
    """configs[4]: 8 x 4 GiB.  Every shard but the first has buf_global_off > 2^32."""
    from fuzzysearch_amd import _native
    n_got, n_edge = _run(_native.Engine, 8, 4096)
    assert n_got > 1000 and n_edge == 3 * 4 + 2 * 3
