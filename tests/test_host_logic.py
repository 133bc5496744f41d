"""Host-side logic of the drop-in layer: parameter validation / normalisation, strategy choice,
Match semantics, input encoding, and libfzhip's host-only consolidation entry points
(fz_consolidate / fz_group_best run on the CPU, no device needed).  CPU only."""
import random

import attr
import pytest

import fuzzysearch_amd as fa
import oracle
from fuzzysearch_amd import _native, common, engine
from fuzzysearch_amd.common import LevenshteinSearchParams, Match
from tests import golden_io


def test_library_loads_and_exports_the_declared_abi():
    import os
    import re
    lib = _native.load_library()
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fzhip.h")).read()
    declared = set(re.findall(r"\b(fz_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_native.EXPORTED_SYMBOLS), declared ^ set(_native.EXPORTED_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.fz_abi_version() == 1


def test_no_device_fails_loudly():
    """Without an MI355X the product raises; it never computes matches on the CPU."""
    import ctypes
    lib = _native.load_library()
    n = ctypes.c_int(-1)
    lib.fz_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_native.HipEngineError):
        fa.find_near_matches(b"PATTERN", b"---PATERN---", max_l_dist=1)


def test_params_validation_and_normalisation():
    """Golden: every LevenshteinSearchParams outcome the reference's dispatcher tests rely on
    (common.py:61-116)."""
    P = LevenshteinSearchParams
    with pytest.raises(ValueError):
        P()
    for bad in [dict(max_substitutions=1), dict(max_insertions=1, max_deletions=1),
                dict(max_substitutions=1, max_insertions=1), dict(max_substitutions=1, max_deletions=1)]:
        with pytest.raises(ValueError):
            P(**bad)
    for bad in [dict(max_l_dist=-1), dict(max_l_dist=1.5), dict(max_substitutions="1", max_l_dist=2)]:
        with pytest.raises(TypeError):
            P(**bad)
    assert P(None, None, None, 2).unpacked == (2, 2, 2, 2)
    assert P(1, 2, 3, None).unpacked == (1, 2, 3, 6)
    assert P(5, 2, 2, 5).unpacked == (5, 2, 2, 5)
    assert P(1, 1, 1, 10).unpacked == (1, 1, 1, 3)
    assert P(3, 0, 0, None).unpacked == (3, 0, 0, 3)
    assert P(None, 0, 0, 4).unpacked == (4, 0, 0, 4)
    assert attr.evolve(P(1, 2, 3, None), max_l_dist=2).unpacked == (1, 2, 2, 2)


def test_choose_search_class_rules():
    """SURVEY.md A.4 / __init__.py:60-83, incl. trap 4 (max_l_dist alone never reaches Generic)."""
    P = LevenshteinSearchParams
    assert fa.choose_search_class(P(0, 0, 0, 0)) is fa.ExactSearch
    assert fa.choose_search_class(P(3, 3, 3, 0)) is fa.ExactSearch
    assert fa.choose_search_class(P(3, 0, 0, None)) is fa.SubstitutionsOnlySearch
    assert fa.choose_search_class(P(3, 0, 0, 5)) is fa.SubstitutionsOnlySearch
    assert fa.choose_search_class(P(None, None, None, 5)) is fa.LevenshteinSearch
    assert fa.choose_search_class(P(5, 5, 5, 5)) is fa.LevenshteinSearch
    assert fa.choose_search_class(P(5, 2, 2, 5)) is fa.GenericSearch
    assert fa.choose_search_class(P(1, 1, 1, None)) is fa.GenericSearch
    assert fa.LevenshteinSearch.extra_items_for_chunked_search(b"x", P(None, None, None, 3)) == 3
    assert fa.GenericSearch.extra_items_for_chunked_search(b"x", P(5, 4, 2, 5)) == 5
    assert fa.ExactSearch.extra_items_for_chunked_search(b"x", P(0, 0, 0, 0)) == 0
    assert fa.SubstitutionsOnlySearch.extra_items_for_chunked_search(b"x", P(2, 0, 0, 2)) == 0


def test_match_semantics():
    a, b = Match(1, 5, 2, b"abcd"), Match(1, 5, 2, "other")
    assert a == b and hash(a) == hash(b) and len({a, b}) == 1        # `matched` is not identity
    assert sorted([Match(3, 4, 0, "x"), Match(1, 9, 2, "y"), Match(1, 5, 2, "z")])[0].end == 5
    assert attr.evolve(a, start=0).start == 0
    with pytest.raises(attr.exceptions.FrozenInstanceError):
        a.start = 3
    for bad in [(-1, 2, 0, "x"), (3, 2, 0, "x"), (1, 2, -1, "x"), (1, 2, 0, None)]:
        with pytest.raises(ValueError):
            Match(*bad)
    assert repr(Match(3, 9, 1, "PATERN")) == "Match(start=3, end=9, dist=1, matched='PATERN')"


def test_c_match_type_behaves_like_the_attrs_class():
    """csrc/_fzmatch.c's Match (start / end / dist as C integers inside the instance) against the attrs class that states the
    reference's (common.py:15-32): construction and its errors, equality / hash / ordering, frozen, repr, attrs' helpers,
    pickling, copying, weak references, the cyclic collector."""
    import copy
    import gc
    import pickle
    import weakref
    from fuzzysearch_amd import build as fzbuild
    fzbuild.build_match_ext()
    A = common._AttrsMatch
    assert common._fzmatch is not None and Match is common._fzmatch.Match and Match is not A
    assert Match.__name__ == A.__name__ == "Match" and Match.__module__ == A.__module__
    assert [a.name for a in attr.fields(Match)] == ["start", "end", "dist", "matched"] and attr.has(Match)
    rnd = random.Random(5)
    vals = [(rnd.randrange(0, 50), rnd.randrange(0, 50), rnd.randrange(0, 4), rnd.choice([b"x", "y", [1], (2,), None]))
            for _ in range(300)] + [(2 ** 40, 2 ** 62, 2 ** 31, "big")]
    b = Match(True, 3, False, "bools are ints, and are read back as ints")
    assert (b.start, b.end, b.dist) == (1, 3, 0) and type(b.start) is int and b == Match(1, 3, 0, "")
    made = []
    for v in vals + [(-1, 2, 0, "x"), (1, 2, -1, "x"), ("1", 2, 0, "x"), (1, 2.0, 0, "x"), (1, 2, None, "x")]:
        outcomes = []
        for cls in (Match, A):
            try:
                outcomes.append(cls(*v))
            except (ValueError, TypeError) as e:
                outcomes.append((type(e), str(e)))
        c, a = outcomes
        if isinstance(a, tuple):
            assert c == a, v                                          # the same exception with the same message
            continue
        made.append((c, a))
        assert repr(c) == repr(a) and attr.astuple(c) == attr.astuple(a) and attr.asdict(c) == attr.asdict(a)
        assert (c.start, c.end, c.dist) == (a.start, a.end, a.dist) and c.matched is a.matched
        assert type(c.start) is int and type(c.end) is int and type(c.dist) is int
        e = attr.evolve(c, end=c.end + 5, matched="other")
        assert type(e) is Match and (e.start, e.end, e.dist, e.matched) == (c.start, c.end + 5, c.dist, "other")
    assert len(made) > 100
    for (c1, a1), (c2, a2) in zip(made, made[1:] + made[:1]):
        assert (c1 == c2, c1 != c2, c1 < c2, c1 <= c2, c1 > c2, c1 >= c2) == (a1 == a2, a1 != a2, a1 < a2, a1 <= a2, a1 > a2, a1 >= a2)
        assert (hash(c1) == hash(c2)) == (hash(a1) == hash(a2))
    assert sorted(c for c, _ in made) == [Match(a.start, a.end, a.dist, a.matched) for a in sorted(a for _, a in made)]
    assert len({c for c, _ in made}) == len({a for _, a in made})
    m = Match(1, 5, 2, b"abcd")
    assert m != (1, 5, 2) and m != A(1, 5, 2, b"abcd") and not (m == 5)   # the same class only, as attrs compares
    with pytest.raises(TypeError):
        m < A(1, 5, 3, b"x")
    with pytest.raises(TypeError):
        Match(1, 2, 3)
    with pytest.raises(TypeError):
        Match(1, 2, 3, "x", extra=1)
    assert Match(start=1, end=5, dist=2, matched="kw") == m == Match(1, 5, matched="kw", dist=2)
    for action in (lambda: setattr(m, "start", 3), lambda: setattr(m, "other", 3), lambda: delattr(m, "matched")):
        with pytest.raises(attr.exceptions.FrozenInstanceError):
            action()
    with pytest.raises(AttributeError):
        m.other
    assert not hasattr(m, "__dict__")
    for clone in (pickle.loads(pickle.dumps(m)), pickle.loads(pickle.dumps(m, 2)), copy.copy(m), copy.deepcopy(m)):
        assert type(clone) is Match and clone == m and clone.matched == m.matched
    assert weakref.ref(m)() is m
    assert Match.__match_args__ == ("start", "end", "dist", "matched")

    class Sub(Match):                                             # subclasses work as with the attrs class
        def span(self):
            return self.end - self.start

    class SubA(A):
        def span(self):
            return self.end - self.start

    sub, sub_a = Sub(1, 5, 2, "x"), SubA(1, 5, 2, "x")
    assert sub.span() == sub_a.span() == 4 and repr(sub) == repr(sub_a).replace("SubA", "Sub") and sub == Sub(1, 5, 2, "y") and sub != m
    with pytest.raises(attr.exceptions.FrozenInstanceError):
        sub.start = 2
    assert weakref.ref(sub)() is sub and gc.is_tracked(sub)
    many = [Sub(i, i + 1, 0, [i]) for i in range(2000)]
    del many, sub
    gc.collect()
    # only a `matched` that could hold a reference back makes the instance visible to the cyclic collector — and then a
    # cycle through it is collected
    assert not gc.is_tracked(m) and not gc.is_tracked(Match(0, 1, 0, "s")) and gc.is_tracked(Match(0, 1, 0, [1]))
    holder = []
    cyc = Match(0, 1, 0, holder)
    holder.append(cyc)
    probe = weakref.ref(cyc)
    del cyc, holder
    gc.collect()
    assert probe() is None


def test_encode_pair_preserves_comparisons():
    p, t, byteslike = engine.encode_pair(b"abc", bytearray(b"xxabcxx"))
    assert byteslike and bytes(t) == b"xxabcxx"
    p, t, byteslike = engine.encode_pair("abc", "xxabcxx")
    assert not byteslike and (p, t) == (b"abc", b"xxabcxx")
    p, t, _ = engine.encode_pair("a中c", "xxa中cx中é")
    assert len(p) == 3 and len(t) == 8
    assert [t[i] == p[j] for i in range(8) for j in range(3)] == \
        ["xxa中cx中é"[i] == "a中c"[j] for i in range(8) for j in range(3)]
    words_p, words_t = "over a lazy dog".split(), "the big brown fox jumped over the lazy dog".split()
    p, t, _ = engine.encode_pair(words_p, words_t)
    assert [t[i] == p[j] for i in range(len(t)) for j in range(len(p))] == \
        [words_t[i] == words_p[j] for i in range(len(words_t)) for j in range(len(words_p))]
    with pytest.raises(TypeError):
        engine.encode_pair("abc", b"abc")
    with pytest.raises(NotImplementedError):
        engine.encode_pair([str(i) for i in range(300)], ["1", "2"])


def test_fz_consolidate_and_group_best_against_oracle_and_golden():
    rnd = random.Random(9)
    for _ in range(3000):
        n = rnd.randint(0, 25)
        raw = []
        for _ in range(n):
            s = rnd.randint(0, 40)
            e = s + rnd.choice([0, 0, 1, 2, 3, 5, 8])
            raw.append((s, e, rnd.randint(0, 3), rnd.randint(0, 2)))
        best, _hull = oracle.group_best(raw)
        assert [b[:3] for b in _native.group_best(raw)] == [b[:3] for b in best], raw
        assert [b[:3] for b in _native.consolidate(raw)] == oracle.consolidate(raw), raw
    for rec in golden_io.load("consolidate_overlapping_matches"):
        raw = golden_io.triples(list(rec["args"][0]))
        got = [b[:3] for b in _native.consolidate(raw)]
        assert golden_io.equal_modulo_ties(got, golden_io.triples(rec["result"]), raw)
    # fz_group_best keeps the groups ordered by hull instead of testing every group for every match: dense and
    # sparse streams with zero-length matches, many merges, up to 1500 rows
    for trial in range(300):
        n = rnd.choice([5, 50, 300, 1500])
        span = rnd.choice([30, 200, 5000, 100000])
        raw = []
        for _ in range(n):
            s = rnd.randint(0, span)
            raw.append((s, s + rnd.choice([0, 0, 1, 2, 3, 5, 8, 40]), rnd.randint(0, 3), rnd.randint(0, 2)))
        best, _hull = oracle.group_best(raw)
        assert [b[:3] for b in _native.group_best(raw)] == [b[:3] for b in best], (trial, n, span)
    # round 4: the fast form of fz_group_best (components by sort + sweep, every component replays its own members) on
    # what the substitutions-only search emits — block-major runs of equal-length windows with cross-block duplicates —
    # plus mixed lengths, zero-length rows, chains that merge several groups, and streams with one crowded component
    # (> 256 members: the exact walk)
    for trial in range(400):
        n = rnd.choice([32, 40, 200, 2800, 6000])
        span = rnd.choice([300, 5000, 200000, 1 << 34])
        m = rnd.choice([1, 5, 20, 32])
        raw = []
        if trial % 2:
            starts = sorted(rnd.randint(0, span) for _ in range(n // 3))
            for g in range(rnd.randint(1, 6)):                       # block-major: every block finds most windows again
                raw += [(s, s + m, rnd.randint(0, 3), g) for s in starts if rnd.random() < 0.8]
        else:
            for _ in range(n):
                s = rnd.randint(0, span)
                raw.append((s, s + rnd.choice([0, 0, 1, 2, 3, 5, 8, 40]), rnd.randint(0, 3), rnd.randint(0, 2)))
        best, _hull = oracle.group_best(raw)
        assert [b[:3] for b in _native.group_best(raw)] == [b[:3] for b in best], (trial, n, span, m)
    # Large streams: the run-folding pass followed by the slice order of the hulls, with zero-length rows at hull
    # edges, rows in block-major runs like the generic search emits, and shuffled.
    for n, span, run in [(3000, 60000, 1), (2600, 2000000, 1), (4000, 30000, 1), (60000, 4000000, 20), (30000, 90000, 7)]:
        raw = []
        while len(raw) < n:
            s0 = rnd.randint(0, span)
            for _ in range(run):
                s = s0 + (rnd.randint(-4, 4) if run > 1 else 0)
                raw.append((s, s + rnd.choice([0, 0, 1, 2, 5, 9, 30]), rnd.randint(0, 3), rnd.randint(0, 2)))
        want = oracle.consolidate(raw)
        assert [b[:3] for b in _native.consolidate(raw)] == want, (n, span, run)
        rnd.shuffle(raw)
        assert [b[:3] for b in _native.consolidate(raw)] == want, (n, span, run, "shuffled")
    # The stream of an n-gram search: block-major, every block's rows ascending — up to 8 ascending runs of hulls are merged
    # and swept in one pass (consolidate_hulls), more go through the slices.  Zero-length rows, equal starts across and
    # inside blocks, rows that bridge two groups only in a later block.
    for trial in range(120):
        nblocks = rnd.randint(1, 11)
        span = rnd.choice([200, 3000, 100000, 1 << 36])
        starts = sorted(rnd.randint(0, span) for _ in range(rnd.choice([1, 5, 40, 400, 1500])))
        raw = []
        for g in range(nblocks):
            rows = [(s + rnd.choice([0, 0, 0, 1, 3]), rnd.choice([0, 0, 1, 4, 20, 33]), rnd.randint(0, 3)) for s in starts if rnd.random() < 0.7]
            rows.sort(key=lambda r: r[0])
            raw += [(s, s + ln, d, g) for (s, ln, d) in rows]
        assert [b[:3] for b in _native.consolidate(raw)] == oracle.consolidate(raw), (trial, nblocks, span)
        best, _hull = oracle.group_best(raw)
        assert [b[:3] for b in _native.group_best(raw)] == [b[:3] for b in best], (trial, nblocks, span)
    # clustered hulls: nearly all of them in a few crowded slices (the per-slice stable sort), one far outlier
    for n, width, far in [(5000, 400, 1 << 40), (900, 3, 1 << 33), (40, 1000, 1 << 20), (31, 5, 7), (33, 5, 7)]:
        raw = [(far, far + 3, 1, 0)]
        for _ in range(n):
            s0 = rnd.randint(0, width) * 50
            raw.append((s0, s0 + rnd.choice([0, 0, 1, 7, 30, 49, 60]), rnd.randint(0, 3), rnd.randint(0, 2)))
        rnd.shuffle(raw)
        assert [b[:3] for b in _native.consolidate(raw)] == oracle.consolidate(raw), (n, width, far)


def test_match_objects_from_rows_c_extension_equals_python_fill():
    """csrc/_fzmatch.c (built by fuzzysearch_amd.build) against the Python fill of RawMatches._make and against
    Match's own constructor: same objects for bytes / str / list sequences, offsets, empty arrays; bad rows raise."""
    import gc
    import sys
    import numpy as np
    from fuzzysearch_amd import build as fzbuild
    fzbuild.build_match_ext()
    from fuzzysearch_amd import _fzmatch
    assert common._fzmatch is not None, "extension present but not picked up"
    rnd = random.Random(21)
    for seq in (bytes(rnd.randrange(256) for _ in range(500)), "".join(chr(rnd.randrange(32, 1000)) for _ in range(500)),
                [rnd.randrange(10) for _ in range(500)], tuple(range(500)), bytearray(500)):
        for n in (0, 1, 7, 300):
            arr = np.zeros(n, dtype=_native._match_dtype())
            for i in range(n):
                s0 = rnd.randrange(0, 480)
                arr[i] = (s0, s0 + rnd.randrange(0, 20), rnd.randrange(0, 9), rnd.randrange(0, 4))
            for off in (0, 12345678901):
                raw = common.RawMatches(arr, seq, off)
                got = raw._make_from_array(arr)
                want = [Match(int(r['start']) + off, int(r['end']) + off, int(r['dist']), matched=seq[int(r['start']):int(r['end'])])
                        for r in arr]
                assert got == want and raw._make(arr.tolist()) == want
                assert all(type(g) is Match and g.matched == w.matched and type(g.matched) is type(w.matched) and hash(g) == hash(w)
                           for g, w in zip(got, want))
                assert all(type(g.start) is int and type(g.dist) is int for g in got)
    # frozen like any Match, collectable, no reference leaked on the sequence
    seq = bytes(1000)
    arr = np.zeros(50, dtype=_native._match_dtype())
    arr['end'] = 10
    before = sys.getrefcount(seq)
    for _ in range(200):
        ms = _fzmatch.make_matches(arr, seq, 0)
    with pytest.raises(attr.exceptions.FrozenInstanceError):
        ms[0].start = 3
    ms_bytes = ms[:1]
    del ms
    gc.collect()
    assert sys.getrefcount(seq) == before
    bad = np.zeros(3, dtype=_native._match_dtype())
    bad['start'][1] = -1
    with pytest.raises(ValueError):
        _fzmatch.make_matches(bad, seq, 0)
    with pytest.raises(ValueError):
        _fzmatch.make_matches(b"12345", seq, 0)
    with pytest.raises(TypeError):                              # a sequence that cannot be sliced: the error surfaces
        _fzmatch.make_matches(arr, 5, 0)
    # the same rows at a raw address (the result buffer of a C-ABI call: common.matches_from_rows)
    assert _fzmatch.make_matches_at(arr.ctypes.data, len(arr), seq, 7) == [Match(7, 17, 0, seq[0:10])] * 50
    assert _fzmatch.make_matches_at(0, 0, seq, 0) == []
    with pytest.raises(ValueError):
        _fzmatch.make_matches_at(0, 3, seq, 0)
    with pytest.raises(ValueError):
        _fzmatch.make_matches_at(bad.ctypes.data, 3, seq, 0)
    # slices of a mutable sequence are taken at once; instances with such a `matched` are visible to the collector
    ba = bytearray(b"0123456789" * 100)
    got = _fzmatch.make_matches(arr, ba, 0)
    ba[0:10] = b"x" * 10
    assert got[0].matched == bytearray(b"0123456789") and gc.is_tracked(got[0]) and not gc.is_tracked(ms_bytes[0])
    # non-contiguous views go through the Python fill
    wide = np.zeros(20, dtype=_native._match_dtype())
    wide['end'] = 4
    assert common.RawMatches(wide[::2], seq)._make_from_array(wide[::2]) == [Match(0, 4, 0, seq[0:4])] * 10


def test_python_level_consolidation_helpers():
    ms = [Match(22, 34, 0, "a"), Match(2, 14, 1, "b"), Match(3, 15, 2, "c"), Match(40, 41, 0, "d")]
    assert common.consolidate_overlapping_matches(ms) == [Match(2, 14, 1, "b"), Match(22, 34, 0, "a"), Match(40, 41, 0, "d")]
    assert common.best_of_groups_in_discovery_order(ms) == [Match(22, 34, 0, "a"), Match(2, 14, 1, "b"), Match(40, 41, 0, "d")]
    groups = common.group_matches(ms)
    assert [sorted((m.start for m in g)) for g in groups] == [[22], [2, 3], [40]]
    assert common.get_best_match_in_group(groups[1]) == Match(2, 14, 1, "b")
    assert common.consolidate_overlapping_matches([]) == []
    assert common.count_differences_with_maximum(b"abcd", b"abXX", 5) == 2
    assert common.count_differences_with_maximum(b"abcd", b"XXXX", 2) == 2


def test_file_api_chunk_geometry_with_a_recording_search_class():
    """find_near_matches_in_file must reproduce the reference's chunking:
    _chunk_size windows that overlap by len(p) - 1 + extra, offsets re-based, one consolidation."""
    import io
    seen = []

    class Recorder(common.FuzzySearchBase):
        @classmethod
        def search(cls, subsequence, sequence, search_params):
            seen.append(sequence.encode() if isinstance(sequence, str) else bytes(sequence))
            return [Match(0, 1, 0, sequence[:1])] if len(sequence) else []

        @classmethod
        def extra_items_for_chunked_search(cls, subsequence, search_params):
            return 2
    import tempfile
    data = bytes(range(256)) * 2
    orig = fa.choose_search_class
    fa.choose_search_class = lambda params: Recorder
    keep = 4 - 1 + 2
    try:
        with tempfile.TemporaryFile() as f:              # 'rb+' -> binary reader (__init__.py:129-171)
            f.write(data)
            f.seek(0)
            out = fa.find_near_matches_in_file(b"abcd", f, max_l_dist=2, _chunk_size=100)
        pos, exp = 0, []
        while pos < len(data):                           # reference geometry: keep + (100 - keep) new bytes
            exp.append(data[pos:pos + 100])
            if pos + 100 >= len(data):
                break
            pos += 100 - keep
        assert seen == exp
        assert [m.start for m in out][:3] == [0, 95, 190]
        # no 'b' in mode and not RawIOBase -> the text reader (__init__.py:174-200): keep + 100 new items
        del seen[:]
        text = "".join(chr(65 + i % 26) for i in range(330))
        fa.find_near_matches_in_file("abcd", io.StringIO(text), max_l_dist=2, _chunk_size=100)
        assert seen == [text[0:100].encode(), text[95:200].encode(), text[195:300].encode(), text[295:330].encode()]
    finally:
        fa.choose_search_class = orig
    with pytest.raises(ValueError):
        fa.find_near_matches_in_file(b"", io.BytesIO(data), max_l_dist=1)


def test_wire_format_pack_and_merge_round_trip():
    """fz_wire_pack / fz_wire_merge (host-only C entry points of the multi-rank gather) against a sort:
    random rank streams in reference order, header-only overflow signalling, rejected records."""
    import ctypes
    import numpy as np
    from fuzzysearch_amd import distributed as fzd  # noqa: F401
    from tests import torch_glue
    lib = _native.load_library()
    rnd = random.Random(9)
    H = torch_glue.WIRE_HEADER_ROWS
    for _ in range(100):
        world, nblocks, cap = rnd.randint(1, 6), rnd.randint(1, 5), rnd.choice([4, 16, 64])
        rows = H + cap
        recv = np.zeros((world, rows, 2), dtype=np.int64)
        parts, base = [], 0
        for r in range(world):
            rec = []
            for g in range(nblocks):
                for _i in range(rnd.randint(0, 5)):
                    st = base + rnd.randint(0, 400)
                    rec.append((st, st + rnd.randint(0, 30), rnd.randint(0, 7), g))
            rec.sort(key=lambda x: (x[3], x[0], x[1]))
            arr = torch_glue._as_match_array(rec) if rec else np.empty(0, dtype=torch_glue.MATCH_DTYPE)
            parts.append(rec)
            _native._check(lib.fz_wire_pack(arr.ctypes.data, len(arr), cap, recv[r].ctypes.data))
            base += 1000
        out = np.empty(world * cap, dtype=torch_glue.MATCH_DTYPE)
        total, top = ctypes.c_uint64(), ctypes.c_uint64()
        _native._check(lib.fz_wire_merge(recv.ctypes.data, world, rows, cap, out.ctypes.data, len(out),
                                         ctypes.byref(total), ctypes.byref(top)))
        assert total.value == sum(len(p) for p in parts) and top.value == max(len(p) for p in parts)
        if top.value > cap:
            continue                                              # the caller would re-gather with a larger capacity
        exp = [x for g in range(nblocks) for p in parts for x in p if x[3] == g]
        assert [tuple(x) for x in out[:total.value].tolist()] == exp
    bad = torch_glue._as_match_array([(10, 5, 0, 0)])                    # end < start cannot be encoded
    assert lib.fz_wire_pack(bad.ctypes.data, 1, 4, np.zeros((H + 4, 2), dtype=np.int64).ctypes.data) != 0


def test_scan_launch_plan_separates_block_hashes():
    """fz_debug_launch_plan (host-only): every launch holds at most 8 consecutive blocks whose distinct
    hashes occupy distinct slots of the 32-slot table under the chosen multiplier and slot bits; the
    launches cover all blocks once, in order; equal n-grams may share a launch (and a slot)."""
    import ctypes
    lib = _native.load_library()
    rnd = random.Random(21)

    def le(b, n):
        return int.from_bytes(bytes(b[:n]).ljust(4, b"\0")[:4], "little") & ((1 << (8 * n)) - 1 if n < 4 else 0xffffffff)

    def block_hash(ng, L, k):
        if L <= 4:
            return (le(ng, L) * k) & 0xffffffff
        dh = min(L, 8) - 3
        return ((le(ng[dh:], 3) & 0xffffff) * (k & 0xffffff) + le(ng, 4)) & 0xffffffff

    n_multi = 0
    for _ in range(600):
        sigma = rnd.choice([1, 2, 4, 4, 20, 256])
        alpha = bytes(rnd.sample(range(256), sigma))
        L = rnd.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 16])
        G = rnd.randint(1, 40)
        m = L * G + rnd.randint(0, L - 1)
        p = bytes(rnd.choice(alpha) for _ in range(m))
        out = (ctypes.c_uint32 * (4 * 64))()
        nl = ctypes.c_uint32(0)
        _native._check(lib.fz_debug_launch_plan(p, m, L, out, 64, ctypes.byref(nl)))
        nxt = 0
        for i in range(nl.value):
            g0, nb, hk, shift = out[4 * i:4 * i + 4]
            assert g0 == nxt and 1 <= nb <= 16 and shift in (27, 22, 17, 12, 7, 2) and hk & 1
            slots = {}
            for b in range(g0, g0 + nb):
                h = block_hash(p[b * L:b * L + L], L, hk)
                assert slots.setdefault((h >> shift) & 31, h) == h, (p, L, i)
            nxt = g0 + nb
        assert nxt == m // L
        n_multi += nl.value > -(-(m // L) // 16)
    assert n_multi < 300            # splitting beyond ceil(G / 16) launches is the exception (low-entropy patterns)


def test_record_ordering_host_step():
    """emit_matches (fz_debug_order_records): unordered device records -> rows in (block, hit index) order, empty
    slots dropped; uniform, clustered and single-key distributions, counts around the bucket / std::sort switch."""
    import ctypes
    import numpy as np
    from fuzzysearch_amd import _native
    lib = _native.load_library()
    rec_dt = np.dtype([("key", "<u8"), ("l", "<u4"), ("r", "<u4"), ("dist", "<u4"), ("aux", "<u4")])
    rng = np.random.default_rng(5)
    L = 6
    for n, span, nblocks, clusters in [(0, 10, 1, 0), (1, 10, 1, 0), (2, 5, 3, 0), (63, 1 << 20, 3, 0), (64, 1 << 20, 3, 0),
                                       (2409, 1 << 30, 3, 0), (2409, 1 << 30, 3, 2), (5000, 1 << 40, 300, 3), (40000, 1 << 33, 2, 1),
                                       (3000, 4096, 1, 0), (70000, 1 << 20, 1000, 0)]:
        if clusters:
            centers = rng.integers(0, span, clusters)
            idx = (centers[rng.integers(0, clusters, n)] + rng.integers(0, 5000, n)).astype(np.uint64)
        else:
            idx = rng.integers(0, span, n).astype(np.uint64)
        blk = rng.integers(0, nblocks, n).astype(np.uint64)
        key = (blk << np.uint64(48)) | idx
        key, first = np.unique(key, return_index=True)              # one record per (block, index), as in a search
        order = rng.permutation(len(key))
        key = key[order]
        recs = np.zeros(len(key), dtype=rec_dt)
        recs["key"] = key
        recs["l"] = rng.integers(0, 9, len(key))
        recs["r"] = rng.integers(0, 20, len(key))
        recs["dist"] = rng.integers(0, 3, len(key))
        empty = rng.random(len(key)) < (0.3 if n % 2 else 0.0)
        recs["dist"][empty] = 0xffffffff
        ptr = ctypes.POINTER(_native.FzMatch)()
        cnt = ctypes.c_uint64(0)
        keep = recs[~empty]
        keep = keep[np.argsort(keep["key"], kind="stable")]
        kidx = (keep["key"] & np.uint64((1 << 48) - 1)).astype(np.int64)
        for bounded in (False, True):                            # ranges found by a pass / known from the search
            if bounded:
                _native._check(lib.fz_debug_order_records_bounded(recs.ctypes.data, len(recs), L, max(1, span + 5000), nblocks,
                                                                  ctypes.byref(ptr), ctypes.byref(cnt)))
            else:
                _native._check(lib.fz_debug_order_records(recs.ctypes.data, len(recs), L, ctypes.byref(ptr), ctypes.byref(cnt)))
            got = _native._take_matches_array(lib, ptr, cnt.value)
            assert cnt.value == len(keep)
            assert np.array_equal(got["start"], kidx - keep["l"].astype(np.int64))
            assert np.array_equal(got["end"], kidx + L + keep["r"].astype(np.int64))
            assert np.array_equal(got["dist"], keep["dist"].astype(np.int32))
            assert np.array_equal(got["block"], (keep["key"] >> np.uint64(48)).astype(np.int32))


def test_no_kernel_spills_to_scratch():
    """Compiler resource remarks collected by fuzzysearch_amd/build.py: no kernel may use scratch memory (round 3 caught a 1.5 KB-per-lane copy of the argument struct in every hit-emitting scan
    instance — exact search 3.4x slower, every parity test green)."""
    import os
    import pytest
    from fuzzysearch_amd import build as fzbuild
    if not os.path.exists(fzbuild.RESOURCES):
        pytest.skip("no resource summary next to libfzhip.so (built by an older build.py)")
    rows = {}
    for ln in open(fzbuild.RESOURCES):
        parts = ln.split()
        if len(parts) >= 3:
            rows[parts[0]] = dict(kv.split("=") for kv in parts[1:])
    scan = {k: v for k, v in rows.items() if "fz_scan_kernel" in k}
    assert len(scan) == 100                # round 6: + 4 x 10 instances: bit-vector columns (32-, 64-, 128-bit) and the Hamming count under their queue discipline
    assert len(rows) >= 60 and any("fz_gen_hit_kernel" in k for k in rows) and any("fz_verify_kernel" in k for k in rows)
    for name, r in rows.items():
        # round 4: EVERY kernel (fz_verify_kernel had 232 B of scratch and 57 spilled VGPRs); round 5: the tiled Levenshtein
        # automaton as well (its step's slot form, kept out of the kernel by a compiler bug in round 4, now has a loop shape
        # hipcc compiles correctly: fz_device.h, profiles/r05_levlp_miscompile.txt)
        assert int(r["scratch"]) == 0 and int(r["vgpr_spill"]) == 0, (name, r)
    bits = [v for k, v in rows.items() if "fz_gen_hit_kernelILj1ELb1EE" in k]
    assert len(bits) == 1 and int(bits[0]["vgprs"]) <= 72          # the bit-parallel automaton: one wave per hit, 7 waves per SIMD fit
    headline = [v for k, v in scan.items() if "ILi2ELi3ELb1ELb0ELb1ELi0EE" in k]
    assert len(headline) == 1 and int(headline[0]["occupancy"]) == 7 and int(headline[0]["vgprs"]) <= 72


def test_count_differences_with_maximum_matches_the_reference():
    import random
    from fuzzysearch_amd.common import count_differences_with_maximum as cd
    rnd = random.Random(3)
    for _ in range(300):
        n = rnd.randint(0, 40)
        a = bytes(rnd.choice(b"ab") for _ in range(n))
        b = bytes(rnd.choice(b"ab") for _ in range(n))
        mx = rnd.randint(0, 10)
        want = min(sum(x != y for x, y in zip(a, b)), mx)
        assert cd(a, b, mx) == want and cd(bytearray(a), memoryview(b), mx) == want
        assert cd(list(a), list(b), mx) == (want if mx else sum(x != y for x, y in zip(a, b)))
    import pytest
    with pytest.raises(ValueError):
        cd(b"abc", b"ab", 3)


def test_biopython_seq_inputs_are_unwrapped(monkeypatch):
    """search_exact.py:13-19: a Bio.Seq.Seq is a sequence type when Biopython is installed (it is not here: a stand-in
    module with the same surface)."""
    import sys
    import types
    from fuzzysearch_amd import engine

    class Seq(object):
        def __init__(self, data):
            self._data = bytes(data)

        def __bytes__(self):
            return self._data

        def __len__(self):
            return len(self._data)

        def __getitem__(self, i):
            return Seq(self._data[i]) if isinstance(i, slice) else chr(self._data[i])
    bio, bio_seq = types.ModuleType("Bio"), types.ModuleType("Bio.Seq")
    bio_seq.Seq = Seq
    bio.Seq = bio_seq
    monkeypatch.setitem(sys.modules, "Bio", bio)
    monkeypatch.setitem(sys.modules, "Bio.Seq", bio_seq)
    monkeypatch.setattr(engine, "_BIO_SEQ", [])              # (the lookup is cached; dropped again when the test ends)
    assert engine.encode_pair(b"AC", b"ACAC")[2] is True and engine._bio_seq() is Seq
    p, t, byteslike = engine.encode_pair("ACGT", Seq(b"TTACGTTT"))
    assert (bytes(p), bytes(t), byteslike) == (b"ACGT", b"TTACGTTT", False)
    p, t, byteslike = engine.encode_pair(Seq(b"ACGT"), Seq(b"TTACGTTT"))
    assert (bytes(p), bytes(t), byteslike) == (b"ACGT", b"TTACGTTT", False)
    assert engine.encode_pair(b"ACGT", b"TTACGTTT")[2] is True


def test_sharded_record_ordering_equals_the_global_one():
    """fz_debug_order_segments (per-shard ordering + block-wise merge, what a multi-device / all-gathered search does)
    == fz_debug_order_records on the concatenation, for shards owning ascending index ranges."""
    import ctypes
    import numpy as np
    from fuzzysearch_amd import _native
    lib = _native.load_library()
    rec_dt = np.dtype([("key", "<u8"), ("l", "<u4"), ("r", "<u4"), ("dist", "<u4"), ("aux", "<u4")])
    rng = np.random.default_rng(9)
    for nshards, per, nblocks in [(2, 50, 3), (3, 0, 2), (8, 3000, 3), (8, 9600, 3), (5, 700, 40), (4, 1, 1)]:
        parts, ends, total = [], [], 0
        shard_span = 1 << 32
        for sidx in range(nshards):
            n = int(rng.integers(0, per + 1)) if per else 0
            idx = (sidx * shard_span + rng.integers(0, shard_span, n)).astype(np.uint64)
            key = np.unique((rng.integers(0, nblocks, n).astype(np.uint64) << np.uint64(48)) | idx)
            key = key[rng.permutation(len(key))]
            r = np.zeros(len(key), dtype=rec_dt)
            r["key"] = key
            r["l"] = rng.integers(0, 5, len(key)); r["r"] = rng.integers(0, 9, len(key)); r["dist"] = rng.integers(0, 3, len(key))
            if sidx % 2:
                r["dist"][rng.random(len(key)) < 0.2] = 0xffffffff
            parts.append(r)
            total += len(r)
            ends.append(total)
        recs = np.concatenate(parts) if total else np.zeros(0, dtype=rec_dt)
        ends_a = np.asarray(ends, dtype=np.uint64)

        def call(fn, *args):
            ptr = ctypes.POINTER(_native.FzMatch)()
            cnt = ctypes.c_uint64(0)
            _native._check(fn(*args, ctypes.byref(ptr), ctypes.byref(cnt)))
            return _native._take_matches_array(lib, ptr, cnt.value)
        whole = call(lib.fz_debug_order_records, recs.ctypes.data, len(recs), 6)
        seg = call(lib.fz_debug_order_segments, recs.ctypes.data, ends_a.ctypes.data, nshards, 6)
        assert np.array_equal(whole, seg), (nshards, per, nblocks)


def test_gathered_blocks_parse_regrow_and_merge():
    """fz_debug_gather_merge = the host half of gather_records (what follows the ncclAllGather of a collective search):
    2..8 ranks' [counters][records] blocks -> one stream in the reference's order.  Covered: empty ranks, a rank whose
    count exceeds the capacity (the re-gather decision and its capacity, identical on every rank because it only
    depends on the gathered headers), ranks that are NOT in ownership order (merged by own_lo), empty record slots,
    garbage behind a rank's last record."""
    import ctypes
    import numpy as np
    lib = _native.load_library()
    rec_dt = np.dtype([("key", "<u8"), ("l", "<u4"), ("r", "<u4"), ("dist", "<u4"), ("aux", "<u4")])
    rng = np.random.default_rng(41)
    HDR = 1024
    L = 6

    def run(blocks, world, cap, own_lo):
        ptr = ctypes.POINTER(_native.FzMatch)()
        cnt, need = ctypes.c_uint64(0), ctypes.c_uint64(0)
        lo = None if own_lo is None else np.asarray(own_lo, dtype=np.uint64)
        _native._check(lib.fz_debug_gather_merge(blocks.ctypes.data, world, cap, None if lo is None else lo.ctypes.data, L,
                                                 ctypes.byref(ptr), ctypes.byref(cnt), ctypes.byref(need)))
        if need.value:
            assert not ptr
            return None, need.value
        return _native._take_matches_array(lib, ptr, cnt.value), 0

    def expected(parts):
        allr = np.concatenate(parts) if parts else np.zeros(0, dtype=rec_dt)
        allr = allr[allr["dist"] != 0xffffffff]
        allr = allr[np.argsort(allr["key"], kind="stable")]            # (block << 48 | idx): the reference's order
        idx = (allr["key"] & np.uint64((1 << 48) - 1)).astype(np.int64)
        out = np.zeros(len(allr), dtype=_native._match_dtype())
        out["start"] = idx - allr["l"].astype(np.int64)
        out["end"] = idx + L + allr["r"].astype(np.int64)
        out["dist"] = allr["dist"].astype(np.int32)
        out["block"] = (allr["key"] >> np.uint64(48)).astype(np.int32)
        return out

    cases = 0
    for world in (1, 2, 3, 5, 8):
        for trial in range(12):
            nblocks = int(rng.integers(1, 9))
            span = 1 << 30
            # rank r owns [perm[r] * span, (perm[r] + 1) * span): in rank order for half the trials, shuffled otherwise
            perm = np.arange(world) if trial % 2 == 0 else rng.permutation(world)
            counts = [int(rng.integers(0, 3000)) if rng.random() > 0.25 else 0 for _ in range(world)]
            parts = []
            for r in range(world):
                n = counts[r]
                idx = (int(perm[r]) * span + rng.choice(span, n, replace=False)).astype(np.uint64)
                key = (rng.integers(0, nblocks, n).astype(np.uint64) << np.uint64(48)) | idx
                rec = np.zeros(n, dtype=rec_dt)
                rec["key"] = key[rng.permutation(n)]
                rec["l"] = rng.integers(0, 4, n); rec["r"] = rng.integers(0, 7, n); rec["dist"] = rng.integers(0, 3, n)
                if trial % 3 == 1 and n:
                    rec["dist"][rng.random(n) < 0.15] = 0xffffffff      # slot-per-hit verification: empty slots
                parts.append(rec)
            top = max(counts)
            for cap in sorted({max(1, top // 2), top, top + 17, 4096}):
                blocks = np.zeros((world, HDR + cap * 24), dtype=np.uint8)
                for r in range(world):
                    hdr = blocks[r, :HDR].view("<u8")
                    hdr[1] = counts[r]
                    hdr[0] = 12345                                      # other counters are not the parser's business
                    body = blocks[r, HDR:].view(rec_dt)
                    body[:] = np.frombuffer(rng.bytes(cap * 24), dtype=rec_dt)   # garbage behind the rank's records
                    body[:min(cap, counts[r])] = parts[r][:cap]
                own_lo = [int(perm[r]) * span if counts[r] or trial % 4 else (1 << 64) - 1 for r in range(world)]
                in_order = bool(np.all(perm == np.arange(world)))
                got, need = run(blocks, world, cap, own_lo if not in_order or trial % 4 == 2 else None)
                if top > cap:
                    assert got is None and need == (top + top // 4 + 1023) // 1024 * 1024 and need >= top, (world, cap, top, need)
                else:
                    want = expected(parts)
                    assert need == 0 and np.array_equal(got, want), (world, trial, cap)
                cases += 1
    assert cases > 150


def test_rccl_is_loaded_lazily_and_its_absence_only_disables_the_collectives():
    """libfzhip.so does not link librccl (an install without RCCL must still load and search); fz_comm_* dlopen it on first
    use and return FZ_EUNSUPPORTED when it cannot be had (FZ_NO_RCCL=1 stands in for a missing library)."""
    import os
    import subprocess
    import sys
    out = subprocess.run(["ldd", _native.LIB_PATH], capture_output=True, text=True)
    assert out.returncode == 0 and "librccl" not in out.stdout and "libamdhip64" in out.stdout
    code = ("import ctypes\n"
            "from fuzzysearch_amd import _native\n"
            "L = _native.load_library()\n"
            "buf = ctypes.create_string_buffer(128)\n"
            "rc = L.fz_comm_unique_id(buf, 128)\n"
            "print(rc, L.fz_last_error().decode())\n")
    env = dict(os.environ, FZ_NO_RCCL="1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert res.returncode == 0, res.stderr
    assert res.stdout.startswith("%d RCCL is not available" % _native.FZ_EUNSUPPORTED), res.stdout


def test_scan_regions_partition_tiles_and_workgroups():
    """fz_debug_scan_regions (no device): the regions a scan launch is cut into — the last resident round takes shrinking
    tile shares — must partition the workgroups [0, grid) and the tiles [0, ntiles) in order, shrink monotonically towards
    the end, keep every workgroup's tile iterations inside what a queue code can carry, and fall back to one region
    (n_regions = 0) for launches too small to taper."""
    import ctypes
    import random
    import numpy as np
    from fuzzysearch_amd import _native
    L = _native.load_library()
    tab = np.zeros(4 * 8, dtype=np.uint64)
    n = ctypes.c_uint32(0)

    def regions(ntiles, grid, n_cus, steps=4, fmin=0.25, per_cu=7):
        rc = L.fz_debug_scan_regions(ntiles, grid, n_cus, steps, fmin, per_cu, ctypes.byref(n), ctypes.c_void_p(tab.ctypes.data))
        assert rc == 0
        return [tuple(int(x) for x in tab[4 * r:4 * r + 4]) for r in range(n.value)]

    # the headline launch: 1 GiB = 65 536 tiles, 6 144 workgroups, 256 CUs
    regs = regions(65536, 6144, 256)
    assert len(regs) == 5 and regs[0][:3] == (0, 6144 - 1792, 0) and regs[-1][3] == 65536
    per_wg = [(e - t0) / nw for (_w, nw, t0, e) in regs]
    assert 12.0 < per_wg[0] < 13.0 and all(a > b for a, b in zip(per_wg, per_wg[1:])) and 0.2 < per_wg[-1] / per_wg[0] < 0.3
    rnd = random.Random(11)
    tapered = 0
    for _ in range(3000):
        n_cus = rnd.choice([1, 8, 64, 104, 228, 256, 304])
        grid = rnd.randint(1, 40000)
        ntiles = rnd.randint(grid, grid * rnd.choice([1, 2, 5, 12, 40, 3000]))
        steps, fmin, per_cu = rnd.randint(0, 9), rnd.choice([0.0, 0.05, 0.25, 0.5, 1.0, 3.0]), rnd.randint(1, 12)
        regs = regions(ntiles, grid, n_cus, steps, fmin, per_cu)
        T = n_cus * per_cu
        if steps <= 0 or grid < 2 * T or ntiles < 4 * grid or min(steps, 7) > T:
            assert regs == []
            continue
        if not regs:
            continue                                            # (per-workgroup iterations beyond a queue code: one region)
        tapered += 1
        assert len(regs) == min(steps, 7) + 1
        wg = tile = 0
        for (w0, nw, t0, e) in regs:
            assert (w0, t0) == (wg, tile) and nw > 0 and e >= t0
            assert (e - t0 + nw - 1) // nw < (1 << 14) - 1        # FZ_TITER_MAX
            wg += nw
            tile = e
        assert wg == grid and tile == ntiles and sum(nw for (_w, nw, _t, _e) in regs[1:]) == T
        shares = [(e - t0) / nw for (_w, nw, t0, e) in regs]
        if min(nw for (_w, nw, _t, _e) in regs) >= 64:          # (rounding: the last group takes what is left, within a tile per workgroup)
            assert all(a >= b - 1.0 for a, b in zip(shares, shares[1:]))
    assert tapered > 300


def test_symbol_remapping_of_integer_items_never_wraps():
    """engine._remap_items codes list / tuple inputs for the kernels (every distinct subsequence item -> 1..255, anything else
    -> 0).  Its vectorised integer path must agree with Python's == on every pair of items: uint64 values >= 2**63 next to
    negative values used to be cast to one dtype and wrap (2**64 - 1 coded as -1: a match the reference does not see)."""
    import numpy as np
    rm = engine._remap_items

    def check(sub, seq):
        """The coding is any relabelling that keeps every comparison: code(sub[i]) == code(seq[j]) iff sub[i] == seq[j]
        (and likewise inside the subsequence), nonzero codes for subsequence items."""
        ps, pt = rm(sub, seq)
        sub, seq = list(sub), list(seq)
        assert len(ps) == len(sub) and len(pt) == len(seq) and all(c != 0 for c in ps)
        for i, a in enumerate(sub):
            for j, b in enumerate(sub):
                assert (ps[i] == ps[j]) == bool(a == b), (sub, seq)
            for j, b in enumerate(seq):
                assert (ps[i] == pt[j]) == bool(a == b), (sub, seq, i, j)
        for j, b in enumerate(seq):
            if pt[j] == 0:
                assert all(not (a == b) for a in sub), (sub, seq)
    cases = [([2 ** 64 - 1], [-1, 5]), ([-1], [2 ** 64 - 1, -1]), ([2 ** 64 - 1, 7], [2 ** 64 - 1, 5, 7, 2 ** 63]),
             ([1, 2, 3], [3, 2, 1, 9]), ([2 ** 63], [-2 ** 63, 2 ** 63]), ([0, -5, 5], [5, -5, 0, 2 ** 40]), ([True, 2], [1, 2, 0])]
    for sub, seq in cases:
        check(sub, seq)
    assert rm(np.array([2 ** 64 - 1], dtype=np.uint64), np.array([-1, 5], dtype=np.int64)) == (b"\x01", b"\x00\x00")
    assert rm(np.array([3, 4], dtype=np.uint8), np.array([4, 3, 200], dtype=np.int16)) == (b"\x01\x02", b"\x02\x01\x00")
    rnd = random.Random(4)
    for _ in range(300):
        pool = [rnd.choice([0, 1, -1, 2 ** 63 - 1, 2 ** 63, 2 ** 64 - 1, -2 ** 63, rnd.randint(-9, 9)]) for _ in range(6)]
        sub = [rnd.choice(pool) for _ in range(rnd.randint(1, 5))]
        seq = [rnd.choice(pool) for _ in range(rnd.randint(0, 12))]
        check(sub, seq)


def test_match_loads_pickles_of_the_attrs_form():
    """A pickle made where the C extension is absent (the attrs-style fallback class: copyreg.__newobj__(Match) + a dict or tuple
    state through __setstate__) loads where it is present, and the other way round: results stay portable."""
    import pickle

    def attrs_style(state):
        # protocol 2, as an attrs slots instance of Match pickles: GLOBAL Match, EMPTY_TUPLE, NEWOBJ, <state>, BUILD, STOP
        return b'\x80\x02cfuzzysearch_amd.common\nMatch\n)\x81' + pickle.dumps(state, 2)[2:-1] + b'b.'

    want = Match(3, 9, 1, 'PATERN')
    for state in ({'start': 3, 'end': 9, 'dist': 1, 'matched': 'PATERN'}, (3, 9, 1, 'PATERN')):
        got = pickle.loads(attrs_style(state))
        assert type(got) is Match and got == want and got.matched == 'PATERN' and repr(got) == repr(want)
    assert pickle.loads(pickle.dumps(want)) == want
    blank = Match.__new__(Match)                     # what __newobj__ makes before the state arrives
    assert (blank.start, blank.end, blank.dist, blank.matched) == (0, 0, 0, None)
    with pytest.raises(TypeError):
        blank.__setstate__([1, 2])
