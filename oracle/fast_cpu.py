"""ctypes wrapper of oracle/libkzg_fast_cpu.so (oracle/fast_cpu.cpp): the multi-threaded CPU baseline
(Montgomery limbs, iterative NTT, signed-window Pippenger, OpenMP).  TEST / BENCH INFRASTRUCTURE: only
tests/ and bench.py's cpu_baseline leg may import it; it is never a fallback of the product."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkzg_fast_cpu.so")
CURVE_IDS = {"bn254": 0, "bls12_381": 1}
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE, "libkzg_fast_cpu.so"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "fast_cpu.cpp")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            build()
        L = ctypes.CDLL(_SO)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.fc_fp_limbs.argtypes = [ci]
        L.fc_ntt.argtypes = [ci, vp, sz, vp, ci, ci]
        L.fc_setup.argtypes = [ci, vp, sz, vp, ci]
        L.fc_msm.argtypes = [ci, vp, vp, sz, vp, vp, vp, ci]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _words(v, limbs=4):
    return np.frombuffer(int(v).to_bytes(8 * limbs, "little"), dtype="<u8").copy()


def max_threads():
    return lib().fc_max_threads()


def ntt(curve, data, w, inverse=False, threads=1):
    """data: uint64[n, 4] canonical, n a power of two, transformed in place."""
    assert data.dtype == np.uint64 and data.flags.c_contiguous
    assert lib().fc_ntt(CURVE_IDS[curve], _p(data), data.shape[0], _p(_words(w)), int(inverse), threads) == 0
    return data


def setup(curve, tau, n, threads=1):
    """[tau^i G1], i < n: uint64[n, 2L] canonical affine."""
    L = lib().fc_fp_limbs(CURVE_IDS[curve])
    out = np.zeros((n, 2 * L), dtype=np.uint64)
    assert lib().fc_setup(CURVE_IDS[curve], _p(_words(tau)), n, _p(out), threads) == 0
    return out


def msm(curve, points_xy, scalars, inf=None, threads=1):
    """sum_i scalars[i] * points[i].  Returns (xy uint64[2L], infinity flag)."""
    L = lib().fc_fp_limbs(CURVE_IDS[curve])
    n = scalars.shape[0]
    assert points_xy.shape[0] >= n and scalars.dtype == np.uint64 and points_xy.dtype == np.uint64
    out = np.zeros(2 * L, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    assert lib().fc_msm(CURVE_IDS[curve], _p(points_xy), _p(inf), n, _p(scalars), _p(out), _p(oinf), threads) == 0
    return out, int(oinf[0])
