"""Build libfzhip.so (the gfx950 HIP engine) in-tree with hipcc.

    python -m fuzzysearch_amd.build [--force] [--report]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container; the
resulting fuzzysearch_amd/libfzhip.so travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfzhip.so")
RESOURCES = os.path.join(HERE, "libfzhip.resources.txt")
SOURCES = ["fzhip.hip"]
DEPS = ["fzhip.hip", "fz_kernels.h", "fz_device.h", os.path.join("..", "..", "include", "fzhip.h")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def source_digest():
    """sha256 over the native sources the library is built from (csrc/*, include/fzhip.h): bench.py prints it
    (`csrc_digest`), tests/test_bench_contract.py holds the committed bench line of the round against the tree's."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS + ["_fzmatch.c"]):
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(d.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, report=False):
    if not force and not needs_build():
        return LIB
    # the compiler's per-kernel resource remarks are always collected (they cost nothing): libfzhip.resources.txt next to
    # the library, checked by tests/test_host_logic.py (a kernel that silently starts using scratch memory is a bug)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage"]
    # RCCL (the all-gather of match lists, fz_comm_*) is dlopen-ed on first use (fzhip.hip: rccl_api), not linked: the
    # library loads and searches on an install without librccl; the run path lets that dlopen find ROCm's copy
    rocm_lib = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl", "-Wl,-rpath," + rocm_lib, "-o", LIB + ".tmp"]
    res = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    err = res.stderr.decode("utf-8", "replace")
    remarks = [ln for ln in err.splitlines() if "-Rpass-analysis=kernel-resource-usage" in ln]
    other = [ln for ln in err.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in ln]
    if res.returncode != 0 or any("warning:" in ln or "error:" in ln for ln in other):
        sys.stderr.write("\n".join(other) + "\n")        # (otherwise only the source context of the remarks)
    if res.returncode != 0:
        raise subprocess.CalledProcessError(res.returncode, cmd)
    with open(RESOURCES, "w") as f:
        f.write(kernel_resources_summary(remarks))
    if report:
        sys.stderr.write(open(RESOURCES).read())
    os.replace(LIB + ".tmp", LIB)
    return LIB


def kernel_resources_summary(remarks):
    """One line per kernel: name, VGPRs, scratch bytes per lane, occupancy, SGPR / VGPR spills, static LDS."""
    import re
    rows, cur = [], None
    for ln in remarks:
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", ln)
        if not m:
            continue
        text = m.group(1).strip()
        if text.startswith("Function Name:"):
            cur = {"name": text.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in text:
            k, v = text.rsplit(":", 1)
            cur[k.strip()] = v.strip()
    out = []
    for r in rows:
        out.append("%s vgprs=%s scratch=%s occupancy=%s sgpr_spill=%s vgpr_spill=%s lds=%s" % (
            r["name"], r.get("VGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"),
            r.get("SGPRs Spill", "?"), r.get("VGPRs Spill", "?"), r.get("LDS Size [bytes/block]", "?")))
    return "\n".join(out) + "\n"


def match_ext_path():
    import sysconfig
    return os.path.join(HERE, "_fzmatch" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_match_ext(force=False):
    """The CPython extension that fills Match objects from fz_match rows (csrc/_fzmatch.c; host code, gcc).
    Optional: common.py materialises in Python when it is absent."""
    import sysconfig
    out, src = match_ext_path(), os.path.join(CSRC, "_fzmatch.c")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
    if not cc:
        raise RuntimeError("no C compiler for _fzmatch")
    cmd = [cc, "-O2", "-fPIC", "-shared", "-Wall", "-I" + sysconfig.get_paths()["include"], src, "-o", out + ".tmp"]
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, report="--report" in sys.argv))
    print(build_match_ext(force="--force" in sys.argv))
