"""One headline search (1 GiB DNA, |p| = 20, k = 2) on the library FUZZYSEARCH_HIP_LIB names; prints the kernel time
(device printf of a lab build goes to stdout)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuzzysearch_amd import _native
from tests import workloads
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = _native.Engine([0])
seq, pat, _ = workloads.cfg2(mib << 20, 1024)
h = eng.upload(seq)
for _ in range(3):
    r = eng.lev_ngrams(h, pat.tobytes(), 2, as_array=True)
    print("call:", len(r), "raw matches; kernel ms", eng.kernel_ms(), flush=True)
