"""Experiment: how fast does a file in the page cache reach HBM when the H2D copy reads the mmap'ed pages directly
(no pread into pinned staging)?  python benchmarks/mmap_h2d.py [MiB]"""
import json, mmap, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fuzzysearch_amd import _native
from tests import workloads

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seq = workloads.dna(mib << 20, 3)
d = "/dev/shm" if os.path.isdir("/dev/shm") else None
with tempfile.NamedTemporaryFile(delete=False, dir=d) as f:
    f.write(seq.tobytes())
    name = f.name
eng = _native.default_engine()
try:
    h = eng.upload(seq[:1 << 20]); h.release()
    def timed(label, make):
        best = None
        for _ in range(3):
            buf, closer = make()
            t0 = time.perf_counter()
            h = eng.upload(buf)
            dt = time.perf_counter() - t0
            h.release()
            del buf
            closer()
            best = dt if best is None else min(best, dt)
        print(json.dumps({"case": label, "GB_per_s": round((mib << 20) / best / 1e9, 2), "ms": round(best * 1e3, 2)}), flush=True)
    timed("anonymous numpy array (pageable)", lambda: (seq, lambda: None))
    def mm(flags):
        fd = os.open(name, os.O_RDONLY)
        m = mmap.mmap(fd, 0, flags=flags, prot=mmap.PROT_READ)
        return np.frombuffer(m, dtype=np.uint8), lambda: os.close(fd)
    timed("mmap MAP_SHARED, fresh mapping", lambda: mm(mmap.MAP_SHARED))
    timed("mmap MAP_SHARED | MAP_POPULATE (populate not timed)", lambda: mm(mmap.MAP_SHARED | getattr(mmap, "MAP_POPULATE", 0x8000)))
    def populate_timed():
        fd = os.open(name, os.O_RDONLY)
        t0 = time.perf_counter()
        m = mmap.mmap(fd, 0, flags=mmap.MAP_SHARED | getattr(mmap, "MAP_POPULATE", 0x8000), prot=mmap.PROT_READ)
        dt = time.perf_counter() - t0
        m.close(); os.close(fd)
        return dt
    print(json.dumps({"case": "mmap + MAP_POPULATE alone", "ms": round(min(populate_timed() for _ in range(3)) * 1e3, 2)}), flush=True)
    # batches, as the stream would issue them
    fd = os.open(name, os.O_RDONLY)
    m = mmap.mmap(fd, 0, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
    arr = np.frombuffer(m, dtype=np.uint8)
    step = 64 << 20
    t0 = time.perf_counter()
    for o in range(0, len(arr), step):
        h = eng.upload(arr[o:o + step]); h.release()
    dt = time.perf_counter() - t0
    print(json.dumps({"case": "mmap, 64 MiB uploads back to back", "GB_per_s": round((mib << 20) / dt / 1e9, 2)}), flush=True)
    del arr
    os.close(fd)
finally:
    os.remove(name)
