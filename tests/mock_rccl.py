"""Build / locate the stand-in collective library (tests/mock_rccl.cpp -> tests/libmock_rccl.so).  Test infrastructure:
the product reaches it only when a test points FZ_RCCL_LIB at it (fzhip.hip: rccl_api honours that variable)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "mock_rccl.cpp")
LIB = os.path.join(HERE, "libmock_rccl.so")

# the nine entry points fzhip.hip's rccl_api table binds: the stand-in must export exactly these (plus its marker)
RCCL_ENTRY_POINTS = ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclAllGather",
                     "ncclAllReduce", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-O2", "-fPIC", "-shared", "-Wall", SRC, "-o", LIB + ".tmp", "-lrt"]
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def env(extra=None):
    """Environment of a subprocess whose libfzhip.so should load the stand-in instead of librccl."""
    e = dict(os.environ)
    e["FZ_RCCL_LIB"] = build()
    e.setdefault("FZ_MOCK_RCCL_TIMEOUT_S", "90")
    if extra:
        e.update(extra)
    return e


if __name__ == "__main__":
    print(build(force=True))
