"""Model of the reference's find_near_matches_in_file (src/fuzzysearch/__init__.py:86-200) on top of the
oracle: the reference's chunk loops restated literally, every chunk searched as an independent sequence by
the oracle's restatement of the route the reference's strategy class would take.  TEST INFRASTRUCTURE.
Pinned against the real reference by tests/golden/reference_file_calls.jsonl (tests/test_file_model.py)."""
import base64
import json
import os
import zlib

import oracle

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_file_calls.jsonl")


def load():
    out = []
    with open(FIXTURE) as f:
        for line in f:
            r = json.loads(line)
            r["p"] = zlib.decompress(base64.b64decode(r["p"]))
            r["data"] = zlib.decompress(base64.b64decode(r["data"]))
            out.append(r)
    return out


def chunk_bounds(n, chunk_size, keep, text):
    """-> [(start, end)] of the chunks the reference searches (binary: :129-171, text: :174-200)."""
    out = []
    if not text:
        n_read = min(chunk_size, n)
        pos = n_read
        offset, chunk_len = 0, n_read
        while n_read:
            out.append((offset, offset + chunk_len))
            n_keep = min(keep, chunk_len) if keep > 0 else 0
            offset += chunk_len - n_keep
            n_read = min(chunk_size - n_keep, n - pos)
            pos += n_read
            chunk_len = n_keep + n_read
    else:
        pos = min(chunk_size, n)
        offset, chunk_len = 0, pos
        while chunk_len:
            out.append((offset, offset + chunk_len))
            n_keep = min(keep, chunk_len)
            offset += chunk_len - n_keep
            n_read = min(chunk_size, n - pos)
            pos += n_read
            if n_keep:
                chunk_len = n_keep + n_read
                if chunk_len == n_keep:
                    break
            else:
                chunk_len = n_read
    return out


def route(kwargs):
    """-> (kind, k, limits, extra): the strategy class the reference picks and its chunk overlap."""
    from fuzzysearch_amd import (ExactSearch, GenericSearch, LevenshteinSearch, LevenshteinSearchParams,
                                 SubstitutionsOnlySearch, choose_search_class)
    sp = LevenshteinSearchParams(kwargs.get("max_substitutions"), kwargs.get("max_insertions"),
                                 kwargs.get("max_deletions"), kwargs.get("max_l_dist"))
    cls = choose_search_class(sp)
    ms, mi, md, ml = sp.unpacked
    if cls is ExactSearch:
        return "exact", 0, None, 0
    if cls is SubstitutionsOnlySearch:
        return "subs", min(x for x in (ml, ms) if x is not None), None, 0
    if cls is LevenshteinSearch:
        return "lev", ml, None, ml
    assert cls is GenericSearch
    return "generic", ml, (ms, mi, md, ml), max(x for x in (ml, mi) if x is not None)


def search_chunk(kind, k, limits, p, t, text):
    """Raw (start, end, dist, block) stream of one chunk, as search_class.search(subsequence, chunk) yields it."""
    m = len(p)
    exact = lambda: [(i, i + m, 0, 0) for i in oracle.search_exact(p, t)]       # noqa: E731
    if kind == "exact" or k == 0:
        return exact()
    ngram = m // (k + 1) >= 3
    if kind == "subs":
        if not ngram:
            return oracle.subs_lp_raw(p, t, k)
        raw = oracle.subs_ngrams_raw(p, t, k)
        if text:                         # pure-Python path: every window once, sorted by start
            seen, out = set(), []
            for r in raw:
                if r[0] not in seen:
                    seen.add(r[0])
                    out.append(r)
            return sorted(out)
        best, _hull = oracle.group_best(raw)
        return best
    if kind == "lev":
        return oracle.lev_ngrams_raw(p, t, k) if ngram else oracle.lev_lp_raw(p, t, k)
    return oracle.generic_ngrams_raw(p, t, *limits) if ngram else oracle.generic_lp_raw(p, t, *limits)


def file_raw(p, data, kwargs, chunk_size, text):
    """-> (kind, rows): the concatenated per-chunk streams in file coordinates, rows = (start, end, dist, block, chunk)."""
    kind, k, limits, extra = route(kwargs)
    keep = len(p) - 1 + extra
    rows = []
    for j, (a, e) in enumerate(chunk_bounds(len(data), chunk_size, keep, text)):
        for (s, en, d, g) in search_chunk(kind, k, limits, p, data[a:e], text):
            rows.append((s + a, en + a, d, g, j))
    return kind, rows


def file_result(p, data, kwargs, chunk_size, text):
    """-> (final [(start, end, dist)], raw rows): what find_near_matches_in_file returns (canonical ties)."""
    kind, rows = file_raw(p, data, kwargs, chunk_size, text)
    if kind in ("lev", "generic"):
        return oracle.consolidate([r[:4] for r in rows]), rows
    return [r[:3] for r in rows], rows
