"""-m gpu: the drop-in API on the MI355X replayed against the golden vectors recorded from the
reference's own unit tests (tests/golden/reference_calls.jsonl).  Each test mirrors one of the
reference's behavioural base classes: same inputs, expected outputs taken from the
reference run."""
import pytest

import fuzzysearch_amd as fa
import oracle
from fuzzysearch_amd import levenshtein, levenshtein_ngram, search_exact, substitutions_only
from tests import golden_io

pytestmark = pytest.mark.gpu

_EXC = {"ValueError": ValueError, "TypeError": TypeError}


def _matches(result):
    return [(m.start, m.end, m.dist, m.matched) for m in result]


def _expect(rec):
    return [(m.start, m.end, m.dist, m.matched) for m in rec["result"]]


def _route_is_ngram(m, k):
    return k == 0 or m // (k + 1) >= 3


def test_search_exact_golden(engine):
    n = 0
    for rec in golden_io.load("search_exact"):
        if "raises" in rec:
            if rec["raises"] in _EXC and len(rec["args"]) >= 2 and len(rec["args"][0]) == 0:
                with pytest.raises(_EXC[rec["raises"]]):
                    search_exact.search_exact(*rec["args"], **rec["kwargs"])
            continue
        assert list(search_exact.search_exact(*rec["args"], **rec["kwargs"])) == list(rec["result"]), rec["args"][2:]
        n += 1
    assert n >= 300


def test_levenshtein_ngrams_raw_golden(engine):
    n = 0
    for rec in golden_io.load("find_near_matches_levenshtein_ngrams"):
        if "raises" in rec:
            with pytest.raises(_EXC[rec["raises"]]):
                levenshtein_ngram.find_near_matches_levenshtein_ngrams(*rec["args"])
            continue
        got = levenshtein_ngram.find_near_matches_levenshtein_ngrams(*rec["args"])
        assert _matches(got) == _expect(rec), rec["args"]
        n += 1
    assert n >= 30


def test_find_near_matches_levenshtein_golden(engine):
    n = 0
    for rec in golden_io.load("find_near_matches_levenshtein"):
        sub, seq, k = (list(rec["args"]) + [rec["kwargs"].get("max_l_dist")])[:3]
        if "raises" in rec:
            if len(sub) == 0:
                with pytest.raises(ValueError):
                    list(levenshtein.find_near_matches_levenshtein(sub, seq, k))
            continue
        got = levenshtein.find_near_matches_levenshtein(sub, seq, k)    # n-gram or LP route
        assert _matches(got) == _expect(rec), (sub, seq, k)
        n += 1
    assert n >= 50


def test_substitutions_golden(engine):
    n = 0
    for rec in golden_io.load("find_near_matches_substitutions") + golden_io.load("find_near_matches_substitutions_ngrams"):
        sub, seq, k = (list(rec["args"]) + [rec["kwargs"].get("max_substitutions")])[:3]
        fn = getattr(substitutions_only, rec["fn"])
        if "raises" in rec:
            if rec["raises"] == "ValueError":
                with pytest.raises(ValueError):
                    fn(sub, seq, k)
            continue
        ngram_fn = rec["fn"].endswith("ngrams")
        if ngram_fn and len(sub) // (k + 1) == 0:
            continue
        got, exp = _matches(fn(sub, seq, k)), _expect(rec)
        assert len(got) == len(exp), (sub, seq, k)
        for g, e in zip(got, exp):           # group order pinned; in-group ties are hash-seed dependent
            assert g == e or (g[2] == e[2] and g[1] - g[0] == e[1] - e[0]), (sub, seq, k, got, exp)
        # the has_near_match_* twins (substitutions_only.py:18-34, :139-145, :218-233)
        has_fn = getattr(substitutions_only, rec["fn"].replace("find_near_matches", "has_near_match"))
        assert has_fn(sub, seq, k) is (len(exp) > 0), (sub, seq, k)
        if len(sub) // (k + 1) >= 1:
            assert substitutions_only.has_near_match_substitutions_lp(sub, seq, k) is (len(exp) > 0)
        n += 1
    assert n >= 40


def test_generic_golden(engine):
    """find_near_matches_generic_ngrams / find_near_matches_generic on their n-gram route."""
    from fuzzysearch_amd import generic_search
    n = 0
    for rec in golden_io.load("find_near_matches_generic_linear_programming"):
        sub, seq, params = rec["args"]
        if "raises" in rec or len(sub) == 0:
            continue
        got = generic_search.find_near_matches_generic_linear_programming(sub, seq, fa.LevenshteinSearchParams(*params))
        assert _matches(got) == _expect(rec), (sub, seq, params)
    for rec in golden_io.load("find_near_matches_generic_ngrams") + golden_io.load("find_near_matches_generic"):
        sub, seq, params = rec["args"]
        if "raises" in rec or len(sub) == 0:
            continue
        sp = fa.LevenshteinSearchParams(*params)
        l = sp.max_l_dist
        if rec["fn"].endswith("ngrams"):
            if len(sub) // (l + 1) == 0:
                continue
            got = generic_search.find_near_matches_generic_ngrams(sub, seq, sp)
            assert generic_search.has_near_match_generic_ngrams(sub, seq, sp) is (len(got) > 0)
        else:
            got = generic_search.find_near_matches_generic(sub, seq, sp)
        assert _matches(got) == _expect(rec), (sub, seq, params)
        n += 1
    assert n >= 60


def test_find_near_matches_public_api_golden(engine):
    """Every find_near_matches(...) call of the reference's suite whose route is on the GPU."""
    n = 0
    for rec in golden_io.load("find_near_matches"):
        args, kwargs = rec["args"], rec["kwargs"]
        if "raises" in rec:
            if rec["raises"] in _EXC:
                with pytest.raises(_EXC[rec["raises"]]):
                    fa.find_near_matches(*args, **kwargs)
            continue
        # every route of the reference's dispatcher is on the GPU: a refusal (UnsupportedSearch is a
        # NotImplementedError) fails the replay instead of being counted as skipped
        got, exp = _matches(fa.find_near_matches(*args, **kwargs)), _expect(rec)
        params = fa.LevenshteinSearchParams(*(list(args[2:]) + [None] * 4)[:4]) if len(args) > 2 else \
            fa.LevenshteinSearchParams(**kwargs)
        cls = fa.choose_search_class(params)
        if cls is fa.SubstitutionsOnlySearch:
            assert len(got) == len(exp)
            for g, e in zip(got, exp):
                assert g == e or (g[2] == e[2] and g[1] - g[0] == e[1] - e[0]), (args, kwargs)
        elif got != exp:
            sub, seq = args[0], args[1]
            from fuzzysearch_amd.engine import encode_pair
            p, t, _ = encode_pair(sub, seq)
            k = params.max_l_dist
            ngram = len(sub) // (k + 1) >= 3
            if cls is fa.GenericSearch:
                raw = (oracle.generic_ngrams_raw if ngram else oracle.generic_lp_raw)(bytes(p), bytes(t), *params.unpacked)
            else:
                raw = (oracle.lev_ngrams_raw if ngram else oracle.lev_lp_raw)(bytes(p), bytes(t), k)
            assert golden_io.equal_modulo_ties([g[:3] for g in got], [e[:3] for e in exp], raw), (args, kwargs, got, exp)
        n += 1
    assert n >= 100, n


def test_reference_readme_examples(engine):
    assert fa.find_near_matches('PATTERN', '---PATERN---', max_l_dist=1) == [fa.Match(3, 9, 1, 'PATERN')]
    seq = '''GACTAGCACTGTAGGGATAACAATTTCACACAGGTGGACAATTACATTGAAAATCACAGATTGGTCACACACACATTGGACATACATAGAAACACACACACATACATTAGATACGAACATAGAAACACACATTAGACGCGTACATAGACACAAACACATTGACAGGCAGTTCAGATGATGACGCCCGACTGATACTCGCGTAGTCGTGGGAGGCAAGGCACACAGGGGATAGG'''
    sub = 'TGCACTGTAGGGATAACAAT'
    assert fa.find_near_matches(sub, seq, max_l_dist=2) == [fa.Match(3, 24, 1, 'TAGCACTGTAGGGATAACAAT')]
    assert fa.find_near_matches(sub.encode(), seq.encode(), max_l_dist=2) == [fa.Match(3, 24, 1, b'x')]
