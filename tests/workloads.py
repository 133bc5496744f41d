"""Deterministic synthetic workloads shared by tests/ and bench.py."""
import numpy as np

DNA = np.frombuffer(b'ACGT', dtype=np.uint8)
TEXT65 = np.frombuffer(b'ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789 .,', dtype=np.uint8)


def dna(n, seed):
    return DNA[np.random.default_rng(seed).integers(0, 4, n, dtype=np.uint8)]


def text65(n, seed):
    return TEXT65[np.random.default_rng(seed).integers(0, 65, n, dtype=np.uint8)]


def utf8_text(n, seed):
    """65-symbol ASCII mixed with ~5 % 2-byte UTF-8 code points, as bytes (SURVEY.md §8(d) cfg 4)."""
    rng = np.random.default_rng(seed)
    base = text65(n, seed)
    pos = np.flatnonzero(rng.random(n - 1) < 0.025)
    pos = pos[np.diff(np.concatenate([[-2], pos])) > 1]
    base[pos] = 0xC3
    base[pos + 1] = 0xA0 + rng.integers(0, 32, len(pos), dtype=np.uint8)
    return base


def plant_variants(seq, pattern, n_plant, seed, alphabet=DNA):
    """Overwrite `seq` (uint8 array, in place) with n_plant variants of `pattern` at sorted random
    positions at least 64 apart: variant i % 4 in {exact, 1 substitution, 1 deletion, 1 insertion}.
    Returns the list of (position, kind)."""
    rng = np.random.default_rng(seed)
    n, m = len(seq), len(pattern)
    if n < 4 * (m + 64):
        return []
    pos = np.sort(rng.choice(n - 64 - m, size=min(n_plant, (n - 64 - m) // 128), replace=False))
    out, last = [], -10 ** 9
    for i, p0 in enumerate(pos):
        p0 = int(p0)
        if p0 - last < 64 + m:
            continue
        last = p0
        kind = i % 4
        v = bytearray(pattern.tobytes())
        q = int(rng.integers(1, m - 1))
        if kind == 1:
            cur = v[q]
            idx = int(np.where(alphabet == cur)[0][0]) if cur in alphabet else 0
            v[q] = int(alphabet[(idx + 1) % len(alphabet)])
        elif kind == 2:
            del v[q]
        elif kind == 3:
            v.insert(q, int(alphabet[0]))
        seq[p0:p0 + len(v)] = np.frombuffer(bytes(v), dtype=np.uint8)
        out.append((p0, kind))
    return out


def cfg3(n=2 ** 30, n_plant=1024):
    """BASELINE config 3: n bytes over 65 ASCII symbols, |p| = 32, <= 3 substitutions."""
    seq, pattern = text65(n, 3), text65(32, 33)
    planted = plant_variants(seq, pattern, n_plant, 8, TEXT65)
    return seq, pattern, planted


def cfg4(n=2 ** 30, n_plant=1024):
    """BASELINE config 4: n bytes of UTF-8 text (as bytes), |p| = 64, k = 5 (4a) / limits (5, 2, 2, 5) (4b)."""
    seq, pattern = utf8_text(n, 4), utf8_text(64, 44)
    planted = plant_variants(seq, pattern, n_plant, 9, TEXT65)
    return seq, pattern, planted


def cfg2(n=2 ** 30, n_plant=1024):
    """BASELINE config 2: n bytes of iid DNA, |p| = 20, k = 2, planted variants."""
    seq = dna(n, 20250925)
    pattern = dna(20, 1)
    planted = plant_variants(seq, pattern, n_plant, 7)
    return seq, pattern, planted


def boundary_plants(m, k, shard_bytes, world):
    """BASELINE config 5: exact copies of the pattern around every shard boundary b, at
    deltas from {-m-k, ..., +1}.  Odd boundaries: a copy straddling b (a different split per boundary) with a copy
    k bytes before and one k bytes after it; even boundaries: a copy that ends k bytes before b (delta -m-k: its
    right window reaches into the next shard) and one that starts at b + 1 (its left window reaches back).
    Non-overlapping.  -> sorted global start positions."""
    out = []
    for r in range(1, world):
        b = r * shard_bytes
        if r % 2:
            left = 1 + (5 * r) % (m - 1)                     # bytes of the straddling copy left of the boundary
            out += [b - left - k - m, b - left, b - left + m + k]
        else:
            out += [b - m - k, b + 1]
    return sorted(out)


def apply_plants(shard, shard_lo, positions, pattern):
    """Write the part of every planted copy (global start positions) that falls inside shard [shard_lo, +len)."""
    m, n = len(pattern), len(shard)
    for q in positions:
        lo, hi = max(q, shard_lo), min(q + m, shard_lo + n)
        if lo < hi:
            shard[lo - shard_lo:hi - shard_lo] = pattern[lo - q:hi - q]
