"""CPU: (1) the oracle-based model of find_near_matches_in_file (tests/file_model.py) against the golden
records taken from the reference's own file API (1316 calls: the reference's chunk-boundary sweep up to
2^20-byte chunks, random small-chunk cases, binary and text mode); (2) the segment geometry the GPU
kernels use (fz_segment / fz_hit_in_range / fz_verify_lev in fz_device.h, compiled for the host) against
that model, chunk by chunk and bit-exact."""
import ctypes
import os
import random
import subprocess
import tempfile

import pytest

from tests import file_model, golden_io

HERE = os.path.dirname(os.path.abspath(__file__))


def _same_modulo_group_ties(got, exp):
    return len(got) == len(exp) and all(g == e or (g[2] == e[2] and g[1] - g[0] == e[1] - e[0]) for g, e in zip(got, exp))


def test_model_equals_reference_file_api():
    n = {"lev": 0, "subs": 0, "exact": 0, "generic": 0}
    for rec in file_model.load():
        kind = file_model.route(rec["kwargs"])[0]
        got, rows = file_model.file_result(rec["p"], rec["data"], rec["kwargs"], rec["chunk"], rec["text"])
        exp = [tuple(r) for r in rec["result"]]
        if kind in ("lev", "generic"):
            assert golden_io.equal_modulo_ties(got, exp, [r[:3] for r in rows]), (rec["kwargs"], rec["chunk"], rec["text"], got, exp)
        elif kind == "subs":
            assert _same_modulo_group_ties(got, exp), (rec["kwargs"], rec["chunk"], rec["text"], got, exp)
        else:
            assert got == exp, (rec["kwargs"], rec["chunk"], rec["text"])
        n[kind] += 1
    assert all(v >= 80 for v in n.values()), n


def test_chunk_geometry_is_the_uniform_segment_formula():
    """Full reads make the reference's chunks regular: binary chunk j = [j*S, j*S + C), S = C - keep; text
    chunk j = [j*C - keep, (j+1)*C) (chunk 0 from 0); chunk j >= 1 exists iff j*S + post < n — the closed
    form the stream (fz_stream, FzGeom segments) is built on."""
    rnd = random.Random(3)
    for _ in range(4000):
        keep = rnd.randint(0, 40)
        C = rnd.randint(2 * keep + 2, 300)
        n = rnd.choice([0, 1, keep, C - 1, C, C + 1, rnd.randint(0, 2000)])
        for text in (False, True):
            S, pre, post = (C, keep, 0) if text else (C - keep, 0, keep)
            nseg = 0 if n == 0 else (1 if n <= post else (n - post - 1) // S + 1)
            want = [(max(0, j * S - pre), min(n, (j + 1) * S + post)) for j in range(nseg)]
            assert file_model.chunk_bounds(n, C, keep, text) == want, (n, C, keep, text)


class OutRec(ctypes.Structure):
    _fields_ = [("start", ctypes.c_int64), ("end", ctypes.c_int64), ("dist", ctypes.c_int32), ("block", ctypes.c_int32),
                ("seg", ctypes.c_int64)]


@pytest.fixture(scope="module")
def emul():
    out = os.path.join(tempfile.gettempdir(), "fz_hostemul_seg_%d.so" % os.getpid())
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall",
                           os.path.join(HERE, "host_emul.cpp"), "-o", out])
    L = ctypes.CDLL(out)
    L.emul_search_segments.restype = ctypes.c_int64
    L.emul_search_segments.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32,
                                       ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64,
                                       ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(OutRec), ctypes.c_int64]
    yield L
    os.remove(out)


def test_segment_device_logic_equals_the_file_model(emul):
    """Levenshtein n-gram search over a batch of segments, run through the very functions the kernels call,
    including a batch that covers only the segments [j0, j1) and only the bytes they need."""
    rnd = random.Random(8)
    n_cases = 0
    for _ in range(2500):
        alpha = bytes(rnd.sample(range(65, 91), rnd.choice([2, 3, 4])))
        k = rnd.randint(1, 3)
        m = rnd.randint(3 * (k + 1), 3 * (k + 1) + 8)
        n = rnd.randint(0, 700)
        t = bytes(rnd.choice(alpha) for _ in range(n))
        p = bytes(rnd.choice(alpha) for _ in range(m))
        if n > m and rnd.random() < 0.7:
            st = rnd.randint(0, n - m)
            p = bytearray(t[st:st + m])
            for _e in range(rnd.randint(0, k)):
                p[rnd.randrange(len(p))] = rnd.choice(alpha)
            p = bytes(p)
        keep = m - 1 + k
        C = rnd.randint(2 * keep + 2, 2 * keep + 2 + 120)
        text = rnd.random() < 0.4
        S, pre, post = (C, keep, 0) if text else (C - keep, 0, keep)
        _kind, rows = file_model.file_raw(p, t, {"max_l_dist": k}, C, text)
        nseg = len(file_model.chunk_bounds(n, C, keep, text))
        # whole file as one batch, then split into two batches at a random segment
        cut = rnd.randint(0, nseg)
        got = []
        for (j0, j1) in ((0, cut), (cut, nseg)):
            if j0 == j1:
                continue
            lo = max(0, j0 * S - pre)
            hi = n if j1 == nseg else j1 * S + post
            cap = 4096
            out = (OutRec * cap)()
            c = emul.emul_search_segments(p, m, t, n if j1 == nseg else hi, k, S, pre, post, j0, j1, lo, hi - lo, out, cap)
            assert 0 <= c <= cap
            got += [(out[i].start, out[i].end, out[i].dist, out[i].block, out[i].seg) for i in range(c)]
        got.sort(key=lambda r: (r[4], r[3]))             # chunk-major, block-major; index order is kept (stable)
        assert got == rows, (p, t, k, C, text, cut)
        n_cases += 1
    assert n_cases > 2000
