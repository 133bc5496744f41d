"""find_near_matches_in_file throughput (reference chunk geometry, every chunk on the GPU)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuzzysearch_amd as fa
from tests import workloads
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seq = workloads.dna(mib << 20, 3); pat = workloads.dna(20, 1)
workloads.plant_variants(seq, pat, 64 * mib // 64, 5)
with tempfile.NamedTemporaryFile(delete=False) as f:
    f.write(seq.tobytes()); name = f.name
try:
    fa.find_near_matches(pat.tobytes(), seq[:1 << 20].tobytes(), max_l_dist=2)     # engine start-up is not file throughput
    for cs in (1 << 20, 1 << 24):
        with open(name, 'rb') as f:
            t0 = time.perf_counter(); r = fa.find_near_matches_in_file(pat.tobytes(), f, max_l_dist=2, _chunk_size=cs); dt = time.perf_counter() - t0
        print("chunk %d KiB: %d MiB in %.3f s -> %.1f MB/s, %d matches" % (cs >> 10, mib, dt, mib * 1.048576 / dt, len(r)))
    t0 = time.perf_counter(); r2 = fa.find_near_matches(pat.tobytes(), seq.tobytes(), max_l_dist=2); dt = time.perf_counter() - t0
    print("in-memory (upload + search): %.3f s -> %.1f MB/s, %d matches" % (dt, mib * 1.048576 / dt, len(r2)))
finally:
    os.remove(name)
