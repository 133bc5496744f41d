"""-m gpu: find_near_matches_in_file on the MI355X against the reference's file API.
Expected values are the golden records taken from the reference's own find_near_matches_in_file
(tests/golden/reference_file_calls.jsonl: its chunk-boundary sweep up to 2^20-byte chunks, whole and half
chunk size, binary and text mode, plus 700 random small-chunk cases) — not the in-memory oracle: the two
APIs differ.  Ties inside an overlap group are compared the tie-aware way (trap 3)."""
import io
import os

import numpy as np
import pytest

import fuzzysearch_amd as fa
from fuzzysearch_amd import _file_stream
from tests import file_model, golden_io, workloads

pytestmark = pytest.mark.gpu


class NamedBytesIO(io.BytesIO):
    mode = 'rb'


def _same_modulo_group_ties(got, exp):
    return len(got) == len(exp) and all(g == e or (g[2] == e[2] and g[1] - g[0] == e[1] - e[0]) for g, e in zip(got, exp))


def _check(rec, got_matches, data_view):
    kind = file_model.route(rec["kwargs"])[0]
    got = [(m.start, m.end, m.dist) for m in got_matches]
    exp = [tuple(r) for r in rec["result"]]
    if kind in ("lev", "generic"):
        if got != exp:
            _k, rows = file_model.file_raw(rec["p"], rec["data"], rec["kwargs"], rec["chunk"], rec["text"])
            assert golden_io.equal_modulo_ties(got, exp, [r[:3] for r in rows]), (rec["kwargs"], rec["chunk"], rec["text"], got, exp)
    elif kind == "subs":
        assert _same_modulo_group_ties(got, exp), (rec["kwargs"], rec["chunk"], rec["text"], got, exp)
    else:
        assert got == exp, (rec["kwargs"], rec["chunk"], rec["text"])
    for m in got_matches:
        assert m.matched == data_view[m.start:m.end], (rec["kwargs"], m)


def test_file_api_golden_replay(engine, tmp_path):
    """Every recorded call: in-memory binary file (readinto feed), text file (read + encode feed), and for
    every fourth binary record a real file (pread threads in the library)."""
    recs = file_model.load()
    streamed = 0
    fn = tmp_path / "hay.bin"
    for i, rec in enumerate(recs):
        kw = dict(rec["kwargs"])
        if rec["text"]:
            text = rec["data"].decode('latin-1')
            got = fa.find_near_matches_in_file(rec["p"].decode('latin-1'), io.StringIO(text), _chunk_size=rec["chunk"], **kw)
            _check(rec, got, text)
        else:
            got = fa.find_near_matches_in_file(rec["p"], NamedBytesIO(rec["data"]), _chunk_size=rec["chunk"], **kw)
            _check(rec, got, rec["data"])
            if i % 4 == 0:
                fn.write_bytes(rec["data"])
                with open(fn, 'rb') as f:
                    got = fa.find_near_matches_in_file(rec["p"], f, _chunk_size=rec["chunk"], **kw)
                    assert f.tell() == len(rec["data"])
                _check(rec, got, rec["data"])
        sp = fa.LevenshteinSearchParams(kw.get("max_substitutions"), kw.get("max_insertions"), kw.get("max_deletions"), kw.get("max_l_dist"))
        cls = fa.choose_search_class(sp)
        keep = len(rec["p"]) - 1 + cls.extra_items_for_chunked_search(rec["p"], sp)
        sub = rec["p"].decode('latin-1') if rec["text"] else rec["p"]
        if _file_stream.plan(cls, sub, sp, rec["chunk"], keep, not rec["text"], NamedBytesIO(b'')) is not None:
            streamed += 1
    assert len(recs) >= 1300 and streamed >= 900, (len(recs), streamed)


def test_file_api_default_chunks_many_batches(engine, tmp_path):
    """A 200 MiB file with the default 1 MiB chunks (several 64 MiB batches, double-buffered staging),
    variants planted on and around chunk and batch boundaries; expected = the reference's chunk loop run
    chunk by chunk through the oracle (tests/file_model.py, itself pinned to the reference)."""
    n = 200 << 20
    seq = workloads.dna(n, 4242)
    pattern = workloads.dna(20, 1)
    p = pattern.tobytes()
    workloads.plant_variants(seq, pattern, 300, 5)
    C, keep = 1 << 20, 20 - 1 + 2
    S = C - keep
    for j in (1, 2, 63, 64, 65, 128, 199):                 # chunk j starts at j * S: matches straddling it
        for delta in (-25, -21, -20, -10, -1, 0, 1):
            pos = j * S + delta + 40 * (delta + 25)
            if 0 <= pos < n - 20:
                seq[pos:pos + 20] = pattern
        seq[j * S - 12:j * S + 8] = pattern
        seq[j * S + keep - 10:j * S + keep + 10] = pattern
    data = seq.tobytes()
    fn = tmp_path / "big.bin"
    fn.write_bytes(data)
    exp, rows = file_model.file_result(p, data, {"max_l_dist": 2}, C, False)
    with open(fn, 'rb') as f:
        got = fa.find_near_matches_in_file(p, f, max_l_dist=2)
    got_t = [(m.start, m.end, m.dist) for m in got]
    assert got_t == exp or golden_io.equal_modulo_ties(got_t, exp, [r[:3] for r in rows])
    assert len(exp) > 300
    assert all(bytes(m.matched) == data[m.start:m.end] for m in got)
    # substitutions-only and exact on the same file (windows belong to exactly one chunk)
    for kw in ({"max_substitutions": 2, "max_insertions": 0, "max_deletions": 0}, {"max_l_dist": 0}):
        exp, _rows = file_model.file_result(p, data, kw, C, False)
        with open(fn, 'rb') as f:
            got = fa.find_near_matches_in_file(p, f, **kw)
        got_t = [(m.start, m.end, m.dist) for m in got]
        assert _same_modulo_group_ties(got_t, exp), kw


def test_file_and_memory_searches_from_several_threads(engine, tmp_path):
    """The default engine is shared between threads: a file stream owns it from open to finish (advisor finding,
    round 2: a search from another thread used to fail — or corrupt the stream — between two batches).  File
    searches (regular file, in-memory file, text file, gzip wrapper) and in-memory searches run concurrently."""
    import io
    import threading

    class Wrapper(io.BytesIO):
        """A reader whose fileno() is NOT its logical content (like GzipFile / BZ2File / decrypting readers)."""
        mode = 'rb'

        def __init__(self, data, other_fd):
            super().__init__(data)
            self._fd = other_fd

        def fileno(self):
            return self._fd
    n = 24 << 20
    seq = workloads.dna(n, 99)
    pattern = workloads.dna(20, 1)
    p = pattern.tobytes()
    workloads.plant_variants(seq, pattern, 200, 5)
    data = seq.tobytes()
    fn = tmp_path / "t.bin"
    fn.write_bytes(data)
    other = tmp_path / "other.bin"
    other.write_bytes(workloads.dna(n, 5).tobytes())
    text = data.decode('ascii')
    tfn = tmp_path / "t.txt"
    tfn.write_text(text, encoding='ascii')
    want_file = [(m.start, m.end, m.dist) for m in fa.find_near_matches_in_file(p, io.BytesIO(data), max_l_dist=2, _chunk_size=1 << 18)]
    want_mem = [(m.start, m.end, m.dist) for m in fa.find_near_matches(p, data[:4 << 20], max_l_dist=2)]
    assert len(want_file) > 150
    errs, res = [], {}

    def work(i):
        try:
            for rep in range(3):
                if i % 5 == 0:
                    with open(fn, 'rb') as f:
                        got = fa.find_near_matches_in_file(p, f, max_l_dist=2, _chunk_size=1 << 18)
                elif i % 5 == 1:
                    got = fa.find_near_matches_in_file(p, io.BytesIO(data), max_l_dist=2, _chunk_size=1 << 18)
                elif i % 5 == 2:
                    with open(other, 'rb') as o:               # fileno() is ANOTHER regular file: must go through readinto()
                        got = fa.find_near_matches_in_file(p, Wrapper(data, o.fileno()), max_l_dist=2, _chunk_size=1 << 18)
                elif i % 5 == 3:
                    with open(tfn, 'r', encoding='ascii') as f:
                        got = fa.find_near_matches_in_file(p.decode('ascii'), f, max_l_dist=2, _chunk_size=1 << 18)
                    assert all(m.matched == text[m.start:m.end] for m in got)
                else:
                    got = fa.find_near_matches(p, data[:4 << 20], max_l_dist=2)
                    assert [(m.start, m.end, m.dist) for m in got] == want_mem
                    continue
                assert [(m.start, m.end, m.dist) for m in got] == want_file, i
            res[i] = True
        except Exception as e:          # noqa: BLE001
            errs.append((i, repr(e)))
    th = [threading.Thread(target=work, args=(i,)) for i in range(10)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    assert len(res) == 10


def test_text_file_short_reads_and_reread(tmp_path):
    """read(n) may return fewer than n characters before EOF (only '' ends the file), and `matched` of a seekable
    text file is read back from the file instead of being held in memory."""
    import io

    class Dribble(io.StringIO):
        def read(self, n=-1):
            return super().read(min(n, 1000) if n and n > 0 else n)
    seq = workloads.dna(3 << 20, 7)
    pattern = workloads.dna(20, 1)
    workloads.plant_variants(seq, pattern, 64, 5)
    text = seq.tobytes().decode('ascii')
    p = pattern.tobytes().decode('ascii')
    want = [(m.start, m.end, m.dist) for m in fa.find_near_matches(p, text, max_l_dist=2)]
    f = Dribble(text)
    got = fa.find_near_matches_in_file(p, f, max_l_dist=2)
    assert [(m.start, m.end, m.dist) for m in got] == want and len(want) > 30
    assert all(m.matched == text[m.start:m.end] for m in got)
    assert f.read() == ''                                       # left at the end of the file, like the reference
    fn = tmp_path / "u.txt"
    fn.write_text(text.replace('A', 'é'), encoding='utf-8')   # multi-byte encoding: tell() cookies, not offsets
    text2 = text.replace('A', 'é')
    p2 = p.replace('A', 'é')
    with open(fn, 'r', encoding='utf-8') as f2:
        got2 = fa.find_near_matches_in_file(p2, f2, max_l_dist=2, _chunk_size=1 << 16)
        assert f2.read() == ''
    assert [(m.start, m.end, m.dist) for m in got2] == want
    assert all(m.matched == text2[m.start:m.end] for m in got2)
