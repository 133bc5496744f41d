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


def _other_symbol(cur, alphabet):
    idx = int(np.where(alphabet == cur)[0][0]) if cur in alphabet else 0
    return int(alphabet[(idx + 1) % len(alphabet)])


def plant_edits(seq, pattern, n_plant, seed, alphabet, n_edits, kinds=(1, 2, 3)):
    """Overwrite `seq` in place with n_plant variants of `pattern` at sorted random positions at least 64 apart;
    variant i carries n_edits(i) edits at distinct pattern positions (inner positions, applied right to left so
    that they stay distinct), every edit of a kind drawn from `kinds`: 1 substitution (by another symbol of the
    alphabet), 2 deletion, 3 insertion (of a symbol that differs from both neighbours' originals is not
    guaranteed: the distance of a variant is AT MOST its edit count).  -> list of (position, edits)."""
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
        ne = n_edits(i)
        v = bytearray(pattern.tobytes())
        where = sorted((int(x) for x in rng.choice(np.arange(1, m - 1), size=ne, replace=False)), reverse=True)
        for q in where:
            kind = int(kinds[int(rng.integers(0, len(kinds)))])
            if kind == 1:
                v[q] = _other_symbol(v[q], alphabet)
            elif kind == 2:
                del v[q]
            else:
                v.insert(q, int(alphabet[int(rng.integers(0, len(alphabet)))]))
        seq[p0:p0 + len(v)] = np.frombuffer(bytes(v), dtype=np.uint8)
        out.append((p0, ne))
    return out


def cfg3(n=2 ** 30, n_plant=1024):
    """BASELINE config 3: n bytes over 65 ASCII symbols, |p| = 32, <= 3 substitutions,
    n_plant planted variants with i % 4 substitutions (0, 1, 2, 3: all within the budget)."""
    seq, pattern = text65(n, 3), text65(32, 33)
    planted = plant_edits(seq, pattern, n_plant, 8, TEXT65, lambda i: i % 4, kinds=(1,))
    return seq, pattern, planted


def cfg4(n=2 ** 30, n_plant=1024):
    """BASELINE config 4: n bytes of UTF-8 text (as bytes), |p| = 64, k = 5 (4a) / limits
    (5, 2, 2, 5) (4b); n_plant planted variants with i % 6 = 0..5 mixed edits (substitutions, deletions, insertions)."""
    seq, pattern = utf8_text(n, 4), utf8_text(64, 44)
    planted = plant_edits(seq, pattern, n_plant, 9, TEXT65, lambda i: i % 6)
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


def iter_shard_buffers(world, shard_bytes, halo, fill):
    """Build a sharded global sequence of world * shard_bytes bytes shard by shard, never holding more than two
    shards on the host: `fill(r, out)` writes shard r's own bytes into `out` (a uint8 view of shard_bytes bytes).
    Yields (r, buf, buf_global_off, own_lo, own_hi) with buf = shard r extended by `halo` bytes of each neighbour
    (clamped at the ends of the sequence) — the arguments of fz_seq_add_shard / fz_seq_upload_shard.  The yielded
    buffer is only valid until the next iteration."""
    def make(r):
        b = np.empty(shard_bytes + 2 * halo, dtype=np.uint8)
        fill(r, b[halo:halo + shard_bytes])
        return b
    cur = make(0)
    prev_tail = None
    for r in range(world):
        nxt = make(r + 1) if r + 1 < world else None
        lo = halo
        hi = halo + shard_bytes
        if prev_tail is not None:
            cur[:halo] = prev_tail
            lo = 0
        if nxt is not None:
            cur[hi:] = nxt[halo:2 * halo]
            hi += halo
        yield r, cur[lo:hi], r * shard_bytes - (halo - lo), r * shard_bytes, (r + 1) * shard_bytes
        prev_tail = cur[shard_bytes:shard_bytes + halo].copy()
        cur = nxt


def cfg5_fill(shard_bytes, world, pattern, k, threads=8):
    """BASELINE config 5 (configs[4] of BASELINE.json): shard r of the global DNA sequence =
    1 GiB pieces dna(piece, 20250925 + 64 r + i), 1 024 planted variants per GiB (seed 7 + r), plus the copies of the
    pattern around every shard boundary (boundary_plants).  -> (fill(r, out), sorted boundary plant positions).  The
    pieces of a shard are generated by a small thread pool (numpy's generators release the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    m = len(pattern)
    edge = boundary_plants(m, k, shard_bytes, world)
    piece = 1 << 30

    def fill(r, out):
        spans = [(i, lo, min(piece, shard_bytes - lo)) for i, lo in enumerate(range(0, shard_bytes, piece))]

        def one(sp):
            i, lo, n = sp
            out[lo:lo + n] = dna(n, 20250925 + 64 * r + i)
        if len(spans) > 1 and threads > 1:
            with ThreadPoolExecutor(min(threads, len(spans))) as ex:
                list(ex.map(one, spans))
        else:
            for sp in spans:
                one(sp)
        plant_variants(out, pattern, 1024 * max(1, shard_bytes >> 30), 7 + r)
        apply_plants(out, r * shard_bytes, edge, pattern)
    return fill, edge
