"""CPU: the stand-in collective library of tests/test_gpu_mock_rccl.py builds, exports exactly the entry points that
libfzhip.so's collective table binds (fzhip.hip: rccl_api), is recognised by the library as a stand-in, and reports
misuse that would hang RCCL as an error.  No device is touched."""
import ctypes
import os
import re
import subprocess
import sys

from tests import mock_rccl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stand_in_exports_what_the_collective_table_binds():
    lib = mock_rccl.build()
    src = open(os.path.join(ROOT, "fuzzysearch_amd", "csrc", "fzhip.hip")).read()
    bound = re.findall(r'sym\("(nccl\w+)"\)', src)
    assert sorted(bound) == sorted(mock_rccl.RCCL_ENTRY_POINTS) and len(bound) == 9
    L = ctypes.CDLL(lib)
    for name in bound + ["fzmock_rccl", "fzmock_rccl_stats"]:
        getattr(L, name)
    nm = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln and ln.split()[-1].startswith("nccl")}
    assert exported == set(bound)


def test_library_recognises_the_stand_in_and_real_rccl():
    code = "from fuzzysearch_amd import _native; print(_native.Engine.comm_backend())"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=mock_rccl.env(), cwd=ROOT, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "stand-in", out.stderr[-2000:]
    env = dict(os.environ)
    env.pop("FZ_RCCL_LIB", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() in ("rccl", "None"), out.stderr[-2000:]
    env["FZ_NO_RCCL"] = "1"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "None", out.stderr[-2000:]


def test_stand_in_refuses_what_would_hang_rccl():
    L = ctypes.CDLL(mock_rccl.build())
    L.ncclGetErrorString.restype = ctypes.c_char_p
    assert L.ncclGroupEnd() != 0                                   # no matching ncclGroupStart
    comms = (ctypes.c_void_p * 3)()
    devs = (ctypes.c_int * 3)(0, 0, 0)
    assert L.ncclCommInitAll(comms, 3, devs) == 0
    # one rank of a three-rank one-process communicator calls alone: RCCL would wait for the others forever
    L.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.ncclAllGather(None, None, 16, 0, comms[0], None)
    assert rc != 0 and b"hang" in L.ncclGetErrorString(rc)
    assert L.ncclGroupStart() == 0
    assert L.ncclAllGather(None, None, 16, 0, comms[0], None) == 0
    assert L.ncclAllGather(None, None, 16, 0, comms[1], None) == 0
    assert L.ncclGroupEnd() != 0                                   # two of three ranks
    for c in comms:
        assert L.ncclCommDestroy(ctypes.c_void_p(c)) == 0
    uid = ctypes.create_string_buffer(128)
    assert L.ncclGetUniqueId(uid) == 0 and uid.raw.startswith(b"/fzmock_")
