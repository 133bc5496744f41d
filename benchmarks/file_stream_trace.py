"""Where the streaming file pipeline's host time goes (FZ_STREAM_TRACE=1 prints fill / collect / launch / carry)."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FZ_STREAM_TRACE"] = "1"
from fuzzysearch_amd import _file_stream, _native
from tests import workloads
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seq = workloads.dna(mib << 20, 3)
pat = workloads.dna(20, 1)
workloads.plant_variants(seq, pat, mib, 5)
p = pat.tobytes()
d = "/dev/shm" if os.path.isdir("/dev/shm") else None
with tempfile.NamedTemporaryFile(delete=False, dir=d) as f:
    f.write(seq.tobytes()); name = f.name
try:
    eng = _native.default_engine()
    batches = [int(x) << 20 for x in os.environ.get("BATCHES", "64,128,256").split(",")]
    for batch in batches:
        for threads in [int(x) for x in os.environ.get("THREADS", "16,32").split(",")]:
            for rep in range(2):
                st = _native.FileStream(eng, _file_stream.MODE_LEV, p, (0, 0, 0), 2, (1 << 20) - 21, 0, 21, batch)
                fd = os.open(name, os.O_RDONLY)
                t0 = time.perf_counter()
                st.read_fd(fd, 0, threads)
                raw, seg = st.finish()
                dt = time.perf_counter() - t0
                os.close(fd); st.close()
            print(json.dumps({"batch_MiB": batch >> 20, "threads": threads, "GB_per_s": round((mib << 20) / dt / 1e9, 2), "ms": round(dt * 1e3, 2)}), flush=True)
finally:
    os.remove(name)
