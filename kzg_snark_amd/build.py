"""Builds libkzg_mi355x.so (hand-written HIP for gfx950 + the C-ABI) in-tree.

    python -m kzg_snark_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the resulting .so is git-ignored
but travels to the GPU box with the working tree."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# KZG_BUILD_DIR: build a variant elsewhere (A/B timing, with KZG_EXTRA_HIPCC_FLAGS and KZG_MI355X_LIB)
LIBDIR = os.environ.get("KZG_BUILD_DIR") or os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libkzg_mi355x.so")
SOURCES = ["api.hip", "ntt.hip", "msm.hip", "msm_prep.hip", "poly.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fgpu-rdc" if False else "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-Wno-pass-failed"]
FLAGS += os.environ.get("KZG_EXTRA_HIPCC_FLAGS", "").split()


def _deps():
    out = []
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".hip", ".py")):
                out.append(os.path.join(root, f))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


PYCONV = os.path.join(LIBDIR, "_kzg_pyconv.so")


def build_pyconv(force=False, verbose=True):
    """The CPython helper for int <-> limb marshalling (csrc/pyconv.c, plain C, no GPU code)."""
    import sysconfig
    src = os.path.join(CSRC, "pyconv.c")
    if not (force or _stale(PYCONV, [src])):
        return PYCONV
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], src,
           "-o", PYCONV]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return PYCONV


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    build_pyconv(force, verbose)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = _deps()
    objs = []
    jobs = []
    for s in srcs:
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, deps):
            jobs.append([HIPCC, *FLAGS, "-c", os.path.join(CSRC, s), "-o", obj])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
