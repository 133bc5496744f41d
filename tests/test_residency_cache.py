"""Transparent residency (fuzzysearch_amd/engine.py: ResidencyCache): find_near_matches(p, <bytes or str>) — the
reference's own call form, __init__.py:35-57 — keeps the last few immutable sequences in HBM, so the second query against
the same object uploads nothing.  CPU part: the LRU / reference-count logic on a stub engine.  GPU part: same results on
first and repeat calls, mutable types bypass the cache, eviction releases device memory."""
import threading

import numpy as np
import pytest

from fuzzysearch_amd import engine as fzengine


class _StubHandle(object):
    def __init__(self, log, n):
        self.log, self.n, self.released = log, n, False
        log.append(("upload", n))

    def release(self):
        assert not self.released
        self.released = True
        self.log.append(("release", self.n))


class _StubEngine(object):
    def __init__(self):
        self.log = []

    def upload(self, data):
        return _StubHandle(self.log, len(data))


def test_lru_budget_and_reference_counts_on_a_stub_engine():
    eng = _StubEngine()
    c = fzengine.ResidencyCache(budget=300000)
    a, b, d = b"a" * 100000, b"b" * 100000, b"d" * 150000
    ha, ea = c.acquire(eng, a, lambda: a)
    assert ea is not None and c.info()["misses"] == 1
    c.done(ea)
    ha2, ea2 = c.acquire(eng, a, lambda: 1 / 0)              # resident: make_data is not even called
    assert ha2 is ha and ea2 is ea and c.info()["hits"] == 1
    hb, eb = c.acquire(eng, b, lambda: b)
    c.done(eb)
    assert c.info()["entries"] == 2 and c.info()["bytes"] == 200000
    # a third sequence over the budget evicts the least recently used one (b: a was touched later? no — a is older but IN USE)
    hd, ed = c.acquire(eng, d, lambda: d)
    info = c.info()
    assert info["bytes"] <= 300000 and info["evictions"] == 1
    # a was the least recently used entry and is still in use by ea2: evicted from the table, released by its last user
    assert not ha.released and ea.evicted
    c.done(ea2)
    assert ha.released
    assert not hb.released and not hd.released
    c.done(ed)
    # small, oversized and mutable sequences are never cached: the caller owns those handles
    for seq in (b"x" * 100, bytearray(b"y" * 100000), memoryview(b"z" * 100000), np.zeros(100000, np.uint8), b"w" * 400000):
        h, e = c.acquire(eng, seq, lambda s=seq: s)
        assert e is None and not h.released
    # same id, different object cannot happen while the entry lives (strong reference) — but an equal copy is another object
    b2 = bytes(bytearray(b))
    h2, e2 = c.acquire(eng, b2, lambda: b2)
    assert h2 is not hb
    c.done(e2)
    c.clear()
    assert c.info()["entries"] == 0 and c.info()["bytes"] == 0 and hb.released and hd.released and h2.released
    off = fzengine.ResidencyCache(budget=0)
    h, e = off.acquire(eng, a, lambda: a)
    assert e is None


def test_dropped_sequences_leave_the_cache_at_the_next_call():
    """The reference holds nothing once it has returned (__init__.py:35-57): an entry whose object only the cache still
    references can never be hit again and is dropped — handle released — by the next call, whatever the budget."""
    eng = _StubEngine()
    c = fzengine.ResidencyCache(budget=1 << 30)
    a, b = b"a" * 100000, b"b" * 120000
    ha, ea = c.acquire(eng, a, lambda: a)
    c.done(ea)
    hb, eb = c.acquire(eng, b, lambda: b)
    c.done(eb)
    assert c.info()["entries"] == 2 and c.sweep() == 0            # both still referenced by their owners
    del a
    assert not ha.released
    small = b"s" * 10
    hs, es = c.acquire(eng, small, lambda: small)                  # ANY call sweeps, also one the cache does not track
    assert es is None and ha.released and not hb.released
    assert c.info()["entries"] == 1 and c.info()["bytes"] == 120000 and c.info()["orphans_dropped"] == 1
    # an entry in use is never an orphan, even if its only other owner is the searching frame's argument
    hb2, eb2 = c.acquire(eng, b, lambda: b)
    del b
    assert c.sweep() == 0 and not hb.released
    c.done(eb2)
    assert c.sweep() == 1 and hb.released and c.info()["entries"] == 0 and c.info()["bytes"] == 0


def test_room_is_made_before_the_upload_and_a_failed_upload_is_retried_on_an_empty_cache():
    eng = _StubEngine()
    c = fzengine.ResidencyCache(budget=250000)
    a, b, d = b"a" * 100000, b"b" * 100000, b"d" * 100000
    for s_ in (a, b):
        c.done(c.acquire(eng, s_, lambda s_=s_: s_)[1])
    del eng.log[:]
    c.done(c.acquire(eng, d, lambda: d)[1])
    assert eng.log == [("release", 100000), ("upload", 100000)]   # the evicted sequence's memory is free when the new one arrives

    class _Full(_StubEngine):                                      # the device has room for one more upload only after a release
        def __init__(self):
            _StubEngine.__init__(self)
            self.fail = 1

        def upload(self, data):
            if self.fail and not any(k == "release" for k, _n in self.log):
                self.fail -= 1
                raise fzengine._native.HipEngineError("hipMalloc: out of memory")
            return _StubEngine.upload(self, data)
    full = _Full()
    c2 = fzengine.ResidencyCache(budget=1 << 30)
    full.fail = 0
    c2.done(c2.acquire(full, a, lambda: a)[1])
    full.fail = 1
    h, e = c2.acquire(full, b, lambda: b)
    assert e is not None and not h.released and c2.info()["upload_retries"] == 1
    assert [k for k, _n in full.log] == ["upload", "release", "upload"] and c2.info()["entries"] == 1
    c2.done(e)
    # with nothing to give back the failure is the caller's
    empty = fzengine.ResidencyCache(budget=1 << 30)
    never = _Full()
    never.fail = 5
    with pytest.raises(fzengine._native.HipEngineError):
        empty.acquire(never, a, lambda: a)


def test_bypass_for_the_chunked_file_searches():
    eng = _StubEngine()
    c = fzengine.ResidencyCache(budget=1 << 30)
    a = b"a" * 100000
    with c.bypass():
        assert c.bypassed()
        h, e = c.acquire(eng, a, lambda: a)
        assert e is None and c.info()["entries"] == 0
        with c.bypass():
            assert c.bypassed()
        assert c.bypassed()
    assert not c.bypassed()
    seen = []
    t = threading.Thread(target=lambda: seen.append(c.bypassed()))  # per thread
    with c.bypass():
        t.start()
        t.join()
    assert seen == [False]


def test_concurrent_acquire_release_on_a_stub_engine():
    eng = _StubEngine()
    c = fzengine.ResidencyCache(budget=250000)
    seqs = [bytes([65 + i]) * 100000 for i in range(5)]
    errors = []

    def worker(seed):
        rnd = np.random.default_rng(seed)
        try:
            for _ in range(400):
                s = seqs[int(rnd.integers(len(seqs)))]
                h, e = c.acquire(eng, s, lambda s=s: s)
                assert not h.released
                if e is not None:
                    c.done(e)
                else:
                    h.release()
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[:1]
    c.clear()
    ups = sum(1 for k, _n in eng.log if k == "upload")
    rel = sum(1 for k, _n in eng.log if k == "release")
    assert ups == rel and c.info()["bytes"] == 0              # every upload released exactly once


def test_budget_from_the_environment(monkeypatch):
    for raw, want in (("0", 0), ("64M", 64 << 20), ("2G", 2 << 30), ("1.5K", 1536), ("12345", 12345), ("junk", 8 << 30)):
        monkeypatch.setenv("FUZZYSEARCH_HIP_RESIDENT_CACHE", raw)
        assert fzengine.ResidencyCache().budget == want


@pytest.mark.gpu
def test_find_near_matches_on_plain_bytes_uploads_once(engine):
    import fuzzysearch_amd as fa
    import oracle
    from tests import workloads
    cache = fzengine.residency_cache()
    cache.clear()
    old_budget = cache._budget
    try:
        cache.budget = 24 << 20
        seq = workloads.dna(8 << 20, 41)
        pattern = workloads.dna(20, 1)
        workloads.plant_variants(seq, pattern, 128, 5)
        t, p = seq.tobytes(), pattern.tobytes()
        want = oracle.consolidate(oracle.lev_ngrams_raw(p, t, 2))
        i0 = cache.info()
        first = fa.find_near_matches(p, t, max_l_dist=2)
        assert [(m.start, m.end, m.dist) for m in first] == want
        assert all(m.matched == t[m.start:m.end] for m in first)
        i1 = cache.info()
        assert i1["misses"] == i0["misses"] + 1 and i1["entries"] == 1 and i1["bytes"] == len(t)
        for q, kw in ((p, dict(max_l_dist=2)), (p[:12], dict(max_l_dist=1)), (p, dict(max_substitutions=2, max_insertions=0, max_deletions=0)),
                      (p, dict(max_l_dist=0))):
            got = fa.find_near_matches(q, t, **kw)
            assert len(got) > 0
        i2 = cache.info()
        assert i2["misses"] == i1["misses"] and i2["hits"] == i1["hits"] + 4          # nothing uploaded again
        assert [(m.start, m.end, m.dist) for m in fa.find_near_matches(p, t, max_l_dist=2)] == want
        # mutable sequences bypass the cache and see their own current contents
        ba = bytearray(t)
        assert [(m.start, m.end, m.dist) for m in fa.find_near_matches(p, ba, max_l_dist=2)] == want
        ba[:] = b"A" * len(ba)
        assert fa.find_near_matches(p, ba, max_l_dist=2) == []
        arr = np.frombuffer(t, dtype=np.uint8).copy()
        assert [(m.start, m.end, m.dist) for m in fa.find_near_matches(p, arr, max_l_dist=2)] == want
        arr[:] = 65
        assert fa.find_near_matches(p, arr, max_l_dist=2) == []
        assert cache.info()["entries"] == 1 and cache.info()["misses"] == i2["misses"]
        # str (latin-1): cached as well, `matched` is text
        text = t.decode("latin-1")
        ps = p.decode("latin-1")
        r1 = fa.find_near_matches(ps, text, max_l_dist=2)
        r2 = fa.find_near_matches(ps, text, max_l_dist=2)
        assert [(m.start, m.end, m.dist) for m in r1] == want and r1 == r2 and all(isinstance(m.matched, str) for m in r1)
        i3 = cache.info()
        assert i3["entries"] == 2 and i3["hits"] == cache.info()["hits"]
        # wide code points: symbol remapping depends on the subsequence -> never cached, still right
        wide = text[: 1 << 20] + "Δ"
        assert [(m.start, m.end, m.dist) for m in fa.find_near_matches(ps, wide, max_l_dist=2)] == \
            oracle.consolidate(oracle.lev_ngrams_raw(p, t[: 1 << 20] + b"\x00", 2))
        assert cache.info()["entries"] == 2
        # a third and fourth sequence push the oldest out (24 MiB budget, 8 MiB each); evicted memory is released
        t3 = workloads.dna(8 << 20, 42).tobytes()
        t4 = workloads.dna(8 << 20, 43).tobytes()
        fa.find_near_matches(p, t3, max_l_dist=2)
        fa.find_near_matches(p, t4, max_l_dist=2)
        i4 = cache.info()
        assert i4["entries"] == 3 and i4["bytes"] == 24 << 20 and i4["evictions"] >= 1
        assert [(m.start, m.end, m.dist) for m in fa.find_near_matches(p, t, max_l_dist=2)] == want      # evicted -> uploaded again
        assert cache.info()["misses"] == i4["misses"] + 1
        # switched off: every call uploads, results unchanged
        cache.clear()
        cache.budget = 0
        assert [(m.start, m.end, m.dist) for m in fa.find_near_matches(p, t, max_l_dist=2)] == want
        assert cache.info()["entries"] == 0
    finally:
        cache.clear()
        cache._budget = old_budget


@pytest.mark.gpu
def test_a_dropped_sequence_gives_its_device_memory_back_at_the_next_call(engine):
    """upload, `del`, the next call — any call — frees the HBM and the host bytes; the public handles
    cache_info() / cache_clear(); the default budget comes from the device's free memory."""
    import gc
    import fuzzysearch_amd as fa
    from tests import workloads
    cache = fzengine.residency_cache()
    cache.clear()
    old_budget = cache._budget
    try:
        cache.budget = None                                        # default: a quarter of the free device memory, at most 8 GiB
        free0, total = engine.mem_info()
        assert 0 < cache.budget <= min(8 << 30, free0 // 4 + 1) and total >= free0
        p = workloads.dna(20, 1).tobytes()
        big = workloads.dna(256 << 20, 77).tobytes()
        fa.find_near_matches(p, big, max_l_dist=2)
        info = fa.cache_info()
        assert info["entries"] == 1 and info["bytes"] == len(big)
        free1, _t = engine.mem_info()
        assert free0 - free1 >= 200 << 20                          # the sequence is resident
        del big
        gc.collect()
        assert fa.cache_info()["entries"] == 1                     # nothing has looked yet
        fa.find_near_matches(p, b"ACGT" * 10, max_l_dist=2)        # a small, uncached search is enough
        info = fa.cache_info()
        assert info["entries"] == 0 and info["bytes"] == 0 and info["orphans_dropped"] >= 1
        free2, _t = engine.mem_info()
        assert free2 - free1 >= 200 << 20                          # ... and its device memory is back
        # cache_clear()
        other = workloads.dna(16 << 20, 78).tobytes()
        fa.find_near_matches(p, other, max_l_dist=2)
        assert fa.cache_info()["entries"] == 1
        fa.cache_clear()
        assert fa.cache_info()["entries"] == 0
    finally:
        cache.clear()
        cache._budget = old_budget
