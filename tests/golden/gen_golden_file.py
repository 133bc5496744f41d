#!/usr/bin/env python3
"""Generate tests/golden/reference_file_calls.jsonl: inputs and outputs of the REFERENCE's
find_near_matches_in_file (src/fuzzysearch/__init__.py:86-200), all four native extensions bound.

Build-container only (needs /root/reference and oracle/_ref).  Two families of calls:
  * the chunk-boundary sweep of the reference's own test
    (tests/test_find_near_matches_in_file.py:73-152): chunk sizes 100 .. 2^20, a match placed at deltas
    {-len, -len+1, -4, -2, -1, 0, 1} around the first chunk boundary, whole and half chunk size, binary
    and text mode;
  * random inputs over small alphabets with 64..257-item chunks (where the file API and the in-memory API
    differ): Levenshtein, substitutions-only, exact and generic limits, n-gram and
    linear-programming routes, binary and text mode.

    PYTHONHASHSEED=0 python tests/golden/gen_golden_file.py

Nothing is copied from the reference: the fixture holds inputs/outputs only.
"""
import base64
import io
import json
import os
import random
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "reference_file_calls.jsonl")
assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0 (SURVEY.md trap 3)"
sys.path.insert(0, REPO)
from oracle import ref_loader  # noqa: E402

fz = ref_loader.load_reference_package()


def z(b):
    return base64.b64encode(zlib.compress(bytes(b), 9)).decode("ascii")


class NamedBytesIO(io.BytesIO):
    mode = 'rb'


records = []


def call(pattern, data, kwargs, chunk, text):
    if text:
        f = io.StringIO(data.decode('latin-1'))
        p = pattern.decode('latin-1')
    else:
        f = NamedBytesIO(bytes(data))
        p = pattern
    try:
        res = fz.find_near_matches_in_file(p, f, _chunk_size=chunk, **kwargs)
        out = {"result": [[m.start, m.end, m.dist] for m in res]}
    except Exception as exc:          # noqa: BLE001
        out = {"raises": type(exc).__name__}
    rec = {"p": z(pattern), "data": z(data), "kwargs": kwargs, "chunk": chunk, "text": bool(text)}
    rec.update(out)
    records.append(rec)


# ---- the reference's chunk-boundary sweep ---------------------------------------------------------
for needle, hay, k in [(b'PATTERN', b'PATERN', 0), (b'PATTERN', b'PATERN', 1), (b'PATTERN', b'PATERN', 2),
                       (b'PATTERN', b'PATTERN', 0), (b'PATTERNPATTERN', b'PATTERNPATERN', 2)]:
    for chunk_size in (100, 2 ** 10, 2 ** 12, 2 ** 18, 2 ** 20):
        for delta in sorted({-len(needle), -len(needle) + 1, -4, -2, -1, 0, 1}):
            if len(needle) // (k + 1) < 3 and chunk_size > 2 ** 10:
                continue
            data = bytearray(chunk_size + 100)
            data[chunk_size + delta:chunk_size + delta + len(hay)] = hay
            for cs in (chunk_size, chunk_size // 2):
                for text in (False, True):
                    call(needle, data, {"max_l_dist": k}, cs, text)

# ---- random small-chunk cases ---------------------------------------------------------------------
rnd = random.Random(20260925)
for it in range(700):
    sigma = rnd.choice([2, 2, 3, 4])
    alpha = bytes(rnd.sample(range(65, 91), sigma))
    n = rnd.randint(0, 1500)
    data = bytearray(rnd.choice(alpha) for _ in range(n))
    kind = rnd.choice(["lev", "lev", "lev", "subs", "exact", "generic", "lev_lp"])
    if kind == "lev":
        k = rnd.randint(1, 3)
        m = rnd.randint(3 * (k + 1), 3 * (k + 1) + 10)
        kwargs = {"max_l_dist": k}
    elif kind == "lev_lp":
        k = rnd.randint(1, 2)
        m = rnd.randint(k + 1, 3 * (k + 1) - 1)
        kwargs = {"max_l_dist": k}
    elif kind == "subs":
        k = rnd.randint(1, 3)
        m = rnd.randint(3 * (k + 1), 3 * (k + 1) + 10)
        kwargs = {"max_substitutions": k, "max_insertions": 0, "max_deletions": 0}
    elif kind == "exact":
        k = 0
        m = rnd.randint(2, 12)
        kwargs = {"max_l_dist": 0}
    else:
        ms, mi, md = rnd.randint(0, 2), rnd.randint(0, 2), rnd.randint(0, 2)
        k = max(1, min(3, ms + mi + md))
        if mi == 0 and md == 0:
            mi = 1
        m = rnd.randint(3 * (k + 1), 3 * (k + 1) + 8)
        kwargs = {"max_substitutions": ms, "max_insertions": mi, "max_deletions": md, "max_l_dist": k}
    pattern = bytes(rnd.choice(alpha) for _ in range(m))
    for _ in range(rnd.randint(0, 4)):          # plant edited copies, some of them across chunk boundaries
        v = bytearray(pattern)
        for _e in range(rnd.randint(0, max(1, k))):
            q = rnd.randrange(len(v))
            op = rnd.random()
            if op < 0.4:
                v[q] = rnd.choice(alpha)
            elif op < 0.7 and len(v) > 2:
                del v[q]
            else:
                v.insert(q, rnd.choice(alpha))
        if n > len(v):
            st = rnd.randint(0, n - len(v))
            data[st:st + len(v)] = v
    chunk = rnd.randint(64, 257)
    call(pattern, data, kwargs, chunk, rnd.random() < 0.35)

with open(OUT, "w") as f:
    for rec in records:
        f.write(json.dumps(rec, sort_keys=True) + "\n")
print("wrote %d records (%d raise) to %s" % (len(records), sum("raises" in r for r in records), OUT), file=sys.stderr)
