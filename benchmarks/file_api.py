"""find_near_matches_in_file throughput: the streaming pipeline (fz_stream: pinned double-buffered batches,
many chunks per launch) on a file in the page cache, against the per-chunk path of round 1.
    python benchmarks/file_api.py [MiB]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuzzysearch_amd as fa  # noqa: E402
from fuzzysearch_amd import _file_stream, _native  # noqa: E402
from tests import workloads  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seq = workloads.dna(mib << 20, 3)
pat = workloads.dna(20, 1)
workloads.plant_variants(seq, pat, mib, 5)
p = pat.tobytes()
d = "/dev/shm" if os.path.isdir("/dev/shm") else None
with tempfile.NamedTemporaryFile(delete=False, dir=d) as f:
    f.write(seq.tobytes())
    name = f.name
try:
    fa.find_near_matches(p, seq[:1 << 20].tobytes(), max_l_dist=2)     # engine start-up is not file throughput
    rows = []
    for label, kw, cs in (("lev k=2, 1 MiB chunks", {"max_l_dist": 2}, 1 << 20),
                          ("lev k=2, 16 MiB chunks", {"max_l_dist": 2}, 1 << 24),
                          ("subs<=2, 1 MiB chunks", {"max_substitutions": 2, "max_insertions": 0, "max_deletions": 0}, 1 << 20),
                          ("exact, 1 MiB chunks", {"max_l_dist": 0}, 1 << 20)):
        best = None
        for _ in range(3):
            with open(name, 'rb') as f:
                t0 = time.perf_counter()
                r = fa.find_near_matches_in_file(p, f, _chunk_size=cs, **kw)
                dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rows.append({"case": label, "MiB": mib, "seconds": round(best, 4), "GB_per_s": round((mib << 20) / best / 1e9, 2), "matches": len(r)})
        print(json.dumps(rows[-1]), flush=True)
    # feeding threads: the library's pread pool (page cache -> pinned memory)
    eng = _native.default_engine()
    for threads in (8, 16, 32, 48, 64, 96):
        st = _native.FileStream(eng, _file_stream.MODE_LEV, p, (0, 0, 0), 2, (1 << 20) - 21, 0, 21, _file_stream.BATCH_BYTES)
        fd = os.open(name, os.O_RDONLY)
        t0 = time.perf_counter()
        st.read_fd(fd, 0, threads)
        raw, seg = st.finish()
        dt = time.perf_counter() - t0
        os.close(fd)
        st.close()
        print(json.dumps({"case": "stream only, %d pread threads" % threads, "GB_per_s": round((mib << 20) / dt / 1e9, 2), "raw": len(raw)}), flush=True)
    whole = seq.tobytes()
    t0 = time.perf_counter()
    r2 = fa.find_near_matches(p, whole, max_l_dist=2)
    dt = time.perf_counter() - t0
    print(json.dumps({"case": "in-memory API (pageable upload + search)", "GB_per_s": round((mib << 20) / dt / 1e9, 2), "matches": len(r2)}))
finally:
    os.remove(name)
