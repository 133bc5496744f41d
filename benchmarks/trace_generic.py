"""FZ_TRACE=1 host-side time marks of a few configs[3b] generic searches (C-ABI)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native
from tests import workloads
eng = _native.Engine([0])
seq, pat, _ = workloads.cfg4(1 << 30, 1024)
p = pat.tobytes()
h = eng.upload(seq)
for _ in range(30):
    eng.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True)
os.environ["FZ_TRACE"] = "1"
for _ in range(3):
    r = eng.generic_ngrams(h, p, 5, 2, 2, 5, as_array=True)
    print(eng.kernel_ms(), file=sys.stderr)
