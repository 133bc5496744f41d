#!/usr/bin/env python3
"""Generate tests/golden/reference_calls.jsonl by running the REFERENCE's own unit tests.

Build-container only (needs /root/reference and oracle/_ref).  The reference's test-suite is
organised as behavioural base classes bound to each implementation; here every
function of the hot path is wrapped by a recorder *before* the reference's test modules are
imported, the reference's unittest suite is run with all four native extensions bound
(PYTHONHASHSEED=0), and every direct call a test makes — arguments, and the returned value or the
raised exception type — is written out as DATA.  Because the reference's own assertions ran on
exactly those values, each record of a passing test is a golden vector pinned by the reference.

    PYTHONHASHSEED=0 python tests/golden/gen_golden.py

Nothing is copied from the reference: the fixture holds inputs/outputs only.
"""
import base64
import json
import os
import sys
import types
import unittest
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
OUT = os.path.join(HERE, "reference_calls.jsonl")
MAX_SEQ = 1 << 16

assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0 (SURVEY.md trap 3)"
# `tests` must resolve to the reference's test package, `oracle` to ours
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
sys.path.insert(0, REFERENCE)
sys.path.append(REPO)

from oracle import ref_loader  # noqa: E402

fz = ref_loader.load_reference_package()
from fuzzysearch.common import Match, LevenshteinSearchParams  # noqa: E402

records = []
seen = set()


def enc(x):
    if isinstance(x, Match):
        return {"m": [x.start, x.end, x.dist, enc(x.matched)]}
    if isinstance(x, LevenshteinSearchParams):
        return {"params": list(x.unpacked)}
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, (bytes, bytearray, memoryview)):
        tag = "ba" if isinstance(x, bytearray) else "b"
        raw = bytes(x)
        if len(raw) > 256:                 # long (mostly zero-filled chunk buffers): zlib + base64
            return {tag + "z": base64.b64encode(zlib.compress(raw, 9)).decode("ascii")}
        return {tag: raw.decode("latin-1")}
    if isinstance(x, list):
        return {"l": [enc(v) for v in x]}
    if isinstance(x, tuple):
        return {"t": [enc(v) for v in x]}
    if isinstance(x, (set, frozenset)):
        return {"set": sorted((enc(v) for v in x), key=lambda v: json.dumps(v, sort_keys=True))}
    raise TypeError("cannot encode %r" % type(x))


def too_big(args):
    return any(hasattr(a, "__len__") and not isinstance(a, LevenshteinSearchParams) and len(a) > MAX_SEQ for a in args)


def recorder(name, fn):
    def wrapper(*args, **kwargs):
        try:
            res = fn(*args, **kwargs)
            if isinstance(res, types.GeneratorType):
                res = list(res)
            outcome = ("result", res)
        except Exception as exc:          # noqa: BLE001 - recorded, then re-raised
            outcome = ("raises", type(exc).__name__)
            err = exc
        try:
            if not too_big(args):
                rec = {"fn": name, "args": [enc(a) for a in args],
                       "kwargs": {k: enc(v) for k, v in sorted(kwargs.items())}}
                if outcome[0] == "result":
                    rec["result"] = enc(outcome[1])
                else:
                    rec["raises"] = outcome[1]
                key = json.dumps(rec, sort_keys=True)
                if key not in seen:
                    seen.add(key)
                    records.append(rec)
        except TypeError:
            pass                           # mocks, file objects ...: not data
        if outcome[0] == "raises":
            raise err
        return outcome[1]
    wrapper.__name__ = getattr(fn, "__name__", name)
    return wrapper


def wrap(module, attr, name=None):
    setattr(module, attr, recorder(name or attr, getattr(module, attr)))


import fuzzysearch  # noqa: E402
from fuzzysearch import (common, search_exact, levenshtein, levenshtein_ngram, substitutions_only,  # noqa: E402
                         generic_search)

wrap(fuzzysearch, "find_near_matches")
wrap(search_exact, "search_exact")
wrap(levenshtein, "find_near_matches_levenshtein")
wrap(levenshtein_ngram, "find_near_matches_levenshtein_ngrams")
wrap(levenshtein_ngram, "_expand", "expand")
wrap(levenshtein_ngram, "_expand_short", "expand")
wrap(levenshtein_ngram, "_expand_long", "expand")
wrap(substitutions_only, "find_near_matches_substitutions")
wrap(substitutions_only, "find_near_matches_substitutions_ngrams")
wrap(generic_search, "find_near_matches_generic")
wrap(generic_search, "find_near_matches_generic_ngrams")
wrap(generic_search, "find_near_matches_generic_linear_programming")
wrap(common, "group_matches")
wrap(common, "consolidate_overlapping_matches")
wrap(common, "count_differences_with_maximum")

suite = unittest.defaultTestLoader.discover(os.path.join(REFERENCE, "tests"), top_level_dir=REFERENCE)
result = unittest.TextTestRunner(verbosity=0, stream=open(os.devnull, "w")).run(suite)
print("reference tests run: %d, failures %d, errors %d, skipped %d" %
      (result.testsRun, len(result.failures), len(result.errors), len(result.skipped)), file=sys.stderr)
# the one known error is the reference's dead c_find_near_matches_generic_ngrams
assert not result.failures and len(result.errors) <= 1, (result.failures, result.errors)

# tests/test_find_near_matches.py:12-51 patches the strategy classes with mocks that answer
# [Match(42, 43, 0, 'x')]: those calls exercise the dispatcher, not a search -> not golden data
MOCK = {"l": [{"m": [42, 43, 0, "x"]}]}
records = [r for r in records if not (r["fn"] == "find_near_matches" and r.get("result") == MOCK)]

with open(OUT, "w") as f:
    for rec in records:
        f.write(json.dumps(rec, sort_keys=True) + "\n")
by_fn = {}
for rec in records:
    by_fn[rec["fn"]] = by_fn.get(rec["fn"], 0) + 1
print("wrote %d records to %s: %s" % (len(records), OUT, by_fn), file=sys.stderr)
