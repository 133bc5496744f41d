"""Load the REFERENCE (taleinat/fuzzysearch 0.8.1) for validation.  TEST INFRASTRUCTURE ONLY.

Two levels, because ``/root/reference`` exists only in the build container, never on the GPU box:

* ``load_reference_package()`` -- build container only.  Imports the reference's pure-Python
  package from ``/root/reference/src`` *with all four native extensions bound* (SURVEY.md trap 1:
  "the reference's own C/Cython path" means all four).  The extensions are the ``.so`` files that
  ``oracle/Makefile`` compiled from the reference's own C sources into ``oracle/_ref/``; they are
  pre-registered in ``sys.modules`` under ``fuzzysearch._common`` etc. so that the reference's
  import-time ``try: from fuzzysearch._common import ...`` hooks (search_exact.py:59-77,
  common.py:128-142, levenshtein_ngram.py:146-156, substitutions_only.py:236-285,
  generic_search.py:180-195) pick them up.  Nothing is copied out of ``/root/reference``.

* ``load_reference_natives()`` -- works anywhere ``oracle/_ref/*.so`` travelled to (GPU box too).
  Returns only the natives that do not import the ``fuzzysearch`` package at module init
  (``_common``, ``_substitutions_only``, ``_levenshtein_ngrams``); ``bench.py`` drives these with
  its own restatement of the n-gram loop for the ``cpu_baseline`` leg.
"""
import importlib.machinery
import importlib.util
import os
import sys
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")
REFERENCE_SRC = "/root/reference/src"
_EXT = sysconfig.get_config_var("EXT_SUFFIX")


def _so(name):
    return os.path.join(REF_DIR, name + _EXT)


def have_ref_natives():
    return all(os.path.exists(_so(n)) for n in ("_common", "_substitutions_only", "_levenshtein_ngrams"))


def have_reference_package():
    return os.path.isdir(os.path.join(REFERENCE_SRC, "fuzzysearch")) and have_ref_natives()


def _load_ext(modname, filename):
    loader = importlib.machinery.ExtensionFileLoader(modname, filename)
    spec = importlib.util.spec_from_file_location(modname, filename, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def load_reference_natives():
    """-> dict of the reference's stand-alone native modules (no ``fuzzysearch`` package needed)."""
    if not have_ref_natives():
        raise RuntimeError("oracle/_ref is not built: run `make -C oracle ref` in the build container")
    out = {}
    for name in ("_common", "_substitutions_only", "_levenshtein_ngrams"):
        out[name] = _load_ext("fzref." + name, _so(name))   # PyInit_<last dotted part>
    return out


_pkg = None


def load_reference_package():
    """-> the reference ``fuzzysearch`` package with all four extensions bound (container only)."""
    global _pkg
    if _pkg is not None:
        return _pkg
    if not have_reference_package():
        raise RuntimeError("reference package or oracle/_ref not available")
    if "fuzzysearch" in sys.modules:
        raise RuntimeError("a module named fuzzysearch is already imported")
    # A meta-path finder maps fuzzysearch._<ext> to oracle/_ref/_<ext>.so, so the reference's own
    # import-time hooks find the natives exactly as if they had been built in place.
    class _RefExtFinder(object):
        names = ("_common", "_substitutions_only", "_levenshtein_ngrams", "_generic_search")

        @classmethod
        def find_spec(cls, fullname, path=None, target=None):
            if fullname.startswith("fuzzysearch."):
                short = fullname[len("fuzzysearch."):]
                if short in cls.names and os.path.exists(_so(short)):
                    loader = importlib.machinery.ExtensionFileLoader(fullname, _so(short))
                    return importlib.util.spec_from_file_location(fullname, _so(short), loader=loader)
            return None

    sys.meta_path.insert(0, _RefExtFinder)
    sys.path.insert(0, REFERENCE_SRC)
    try:
        import fuzzysearch  # noqa: the reference
    finally:
        sys.path.remove(REFERENCE_SRC)
    # sanity: the native hooks are really bound
    from fuzzysearch import levenshtein_ngram, search_exact, substitutions_only, generic_search
    assert levenshtein_ngram._expand_short.__name__ == "c_expand_short"
    assert hasattr(search_exact, "_search_exact")
    assert hasattr(substitutions_only, "py_find_near_matches_substitutions_ngrams")
    assert hasattr(generic_search, "c_fnm_generic_lp")
    _pkg = fuzzysearch
    return fuzzysearch
