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


# What the library was built FROM decides whether it is rebuilt, not file times (a checkout, a copy to the GPU box or
# an edited-and-reverted file all move mtimes): sha256 over every source / header of csrc and include, the compiler
# flags and the compiler's version line, kept beside the objects in build_stamp.json.
STAMP = os.path.join(LIBDIR, "build_stamp.json")
LAST = {}          # what the last build() call did: {"mode": "compiled" | "reused", "compiled": [...], "linked": bool}


def _hipcc_version():
    try:
        out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True, timeout=60).stdout
        return next((ln for ln in out.splitlines() if "HIP version" in ln or "clang version" in ln), out[:80]).strip()
    except (OSError, subprocess.SubprocessError):
        return "unknown"


def source_hash(unit=None):
    """Hash of what `unit` (a .hip file of SOURCES; None: the whole library) is compiled from: its own text, every
    header of csrc/ and include/, the flags and the compiler."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(_deps()):
        base = os.path.basename(path)
        if base.endswith(".py") or (base.endswith(".hip") and unit is not None and base != unit):
            continue
        h.update(base.encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(_hipcc_version().encode())
    return h.hexdigest()


def _read_stamp():
    import json
    try:
        with open(STAMP) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


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
    import json
    os.makedirs(LIBDIR, exist_ok=True)
    try:
        build_pyconv(force, verbose)
    except (OSError, subprocess.SubprocessError) as e:
        # optional: _native.ints_to_limbs / limbs_to_ints fall back to int.to_bytes without the helper
        print(f"[build] marshalling helper not built ({e}); the pure-Python conversion is used", flush=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    want = source_hash()
    units = {s: source_hash(s) for s in srcs}
    have = {} if force else _read_stamp().get("units", {})
    objs = []
    jobs = []
    for s in srcs:
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if not (have.get(s) == units[s] and os.path.exists(obj)):
            jobs.append([HIPCC, *FLAGS, "-c", os.path.join(CSRC, s), "-o", obj])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    link = bool(jobs) or force or not os.path.exists(LIB) or _read_stamp().get("source_hash") != want
    if link:
        run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", LIB])
        with open(STAMP, "w") as f:
            json.dump({"source_hash": want, "units": units, "flags": FLAGS, "hipcc": _hipcc_version()}, f, indent=1)
    LAST.clear()
    LAST.update(mode="compiled" if link else "reused", compiled=[os.path.basename(j[-3]) for j in jobs], linked=link,
                source_hash=want)
    if verbose:
        print(f"[build] libkzg_mi355x.so {LAST['mode']}: source hash {want[:16]}, "
              f"{len(jobs)} of {len(srcs)} translation units compiled", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
