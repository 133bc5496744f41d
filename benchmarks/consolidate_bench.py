"""Host-side consolidation on a configs[3b]-shaped raw stream (no GPU needed): 1024 planted sites over 1 GiB,
6 blocks per site, ~34 records per hit in emission (not start) order -> 2.1e5 rows, 1024 results.
Checks fz_consolidate against a plain restatement of the reference's consolidation (common.py:145-189)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fuzzysearch_amd import _native


def raw_stream(seed=3, sites=1024, blocks=6, per_hit=34, n=1 << 30, m=64, k=5):
    rng = np.random.default_rng(seed)
    pos = np.sort(rng.integers(1000, n - 1000, size=sites))
    rows = []
    for g in range(blocks):
        for p in pos:
            s = p + rng.integers(-k, k + 1, size=per_hit)
            e = s + m + rng.integers(-k, k + 1, size=per_hit)
            d = rng.integers(0, k + 1, size=per_hit)
            rows.append(np.stack([s, e, d, np.full(per_hit, g)], axis=1))
    r = np.concatenate(rows)
    arr = np.empty(len(r), dtype=_native._match_dtype())
    arr['start'], arr['end'], arr['dist'], arr['block'] = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    return arr


def reference_consolidate(rows):
    groups = []
    for (s, e, d, b) in rows:
        ov = [g for g in groups if not (e <= g[0] or s >= g[1])]
        if not ov:
            groups.append([s, e, [(s, e, d, b)]])
        else:
            keep = [g for g in groups if g not in ov]
            ms = [(s, e, d, b)]
            for g in ov:
                ms += g[2]
            groups = keep + [[min(x[0] for x in ms), max(x[1] for x in ms), ms]]
    best = [min(g[2], key=lambda x: (x[2], -(x[1] - x[0]), x[0])) for g in groups]
    return sorted(best, key=lambda x: (x[0], x[1], x[2]))


if __name__ == '__main__':
    small = raw_stream(seed=5, sites=40, n=1 << 16)
    got = [tuple(int(v) for v in r)[:3] for r in _native.consolidate_array(small).tolist()]
    want = [x[:3] for x in reference_consolidate([tuple(int(v) for v in r) for r in small.tolist()])]
    assert got == want, (got[:5], want[:5])
    raw = raw_stream()
    for _ in range(3):
        _native.consolidate_array(raw)
    t = []
    for _ in range(20):
        t0 = time.perf_counter()
        out = _native.consolidate_array(raw)
        t.append(time.perf_counter() - t0)
    print({"rows": len(raw), "results": len(out), "best_ms": round(min(t) * 1e3, 3), "median_ms": round(sorted(t)[10] * 1e3, 3)})
