"""Build libcokrige_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python sif-xco2-cokriging_amd/build_native.py [--force]

The shared object lands next to this file so that it travels with the source
tree (the GPU box has no build cache).  No CPU fallback exists: if the library
is missing, importing the numeric entry points raises.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# CK_BUILD_OUT: build an EXPERIMENTAL library next to the product one (own object directory), to be loaded with
# CK_LIB_PATH=<that file> (native.py) -- the product library is never overwritten by an experiment
OUT = os.environ.get("CK_BUILD_OUT") or os.path.join(HERE, "libcokrige_hip.so")
SOURCES = ["ck_api.hip", "ck_cov.hip", "ck_la.hip", "ck_vario.hip", "ck_local.hip", "ck_model.cpp", "ck_host.cpp"]
ARCH = "gfx950"
# ck_vario.hip: the SLP vectoriser packs the binning kernel's per-pair slot counters into 16-bit lanes (v_perm /
# v_pk_add_u16) -- more instructions than the v_addc chain it replaces
EXTRA_FLAGS = {"ck_vario.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "cokrige.h"))
    # experimental builds: one object directory per set of extra definitions (objects are only rebuilt when a SOURCE is
    # newer, so objects compiled with other -D flags must not be picked up)
    defs = os.environ.get("CK_BUILD_DEFS", "") + " " + os.environ.get("CK_EXTRA_HIPCC_FLAGS", "")
    defs = defs.strip()
    if defs and not os.environ.get("CK_BUILD_OUT"):
        # experimental -D flags must never reach the product library (stale objects in build/ would be mixed with them)
        raise RuntimeError("CK_BUILD_DEFS / CK_EXTRA_HIPCC_FLAGS need CK_BUILD_OUT=<path of the experimental library>")
    tag = "" if not defs else "_" + "".join(ch if ch.isalnum() else "_" for ch in defs)
    objdir = os.path.join(HERE, "build" if not os.environ.get("CK_BUILD_OUT") else "build_exp" + tag)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wno-pass-failed"] + os.environ.get("CK_BUILD_DEFS", "").split()
    flags += os.environ.get("CK_EXTRA_HIPCC_FLAGS", "").split()   # kernel experiments (-D...); use with --force

    def compile_one(src):
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        if force or _stale(obj, [path] + headers):
            cmd = [hipcc] + flags + EXTRA_FLAGS.get(src, []) + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-pthread", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
