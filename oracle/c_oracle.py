"""ctypes wrapper of oracle/libkzg_oracle.so (oracle/kzg_oracle.c).  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkzg_oracle.so")
CURVE_IDS = {"bn254": 0, "bls12_381": 1}
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE, "libkzg_oracle.so"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "kzg_oracle.c")):
            build()
        L = ctypes.CDLL(_SO)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        L.oracle_fft.argtypes = [ctypes.c_int, vp, sz, vp, ctypes.c_int]
        L.oracle_fp_limbs.argtypes = [ctypes.c_int]
        L.oracle_g1_mul.argtypes = [ctypes.c_int, vp, ctypes.c_int, vp, vp, vp]
        L.oracle_setup.argtypes = [ctypes.c_int, vp, sz, vp]
        L.oracle_commit.argtypes = [ctypes.c_int, vp, vp, sz, vp, sz, vp, vp]
        L.oracle_open_quotient.argtypes = [ctypes.c_int, vp, vp, sz, sz, vp, vp, vp, vp]
        L.oracle_open_quotient.restype = ctypes.c_long
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _words(v, limbs=4):
    return np.frombuffer(int(v).to_bytes(8 * limbs, "little"), dtype="<u8").copy()


def fft(curve, data, w, inverse=False):
    """data: uint64[n,4] canonical, transformed in place (fft_ff / ifft_ff)."""
    n = data.shape[0]
    rc = lib().oracle_fft(CURVE_IDS[curve], _p(data), n, _p(_words(w)), int(inverse))
    assert rc == 0
    return data


def setup(curve, tau, n):
    L = lib().oracle_fp_limbs(CURVE_IDS[curve])
    out = np.zeros((n, 2 * L), dtype=np.uint64)
    assert lib().oracle_setup(CURVE_IDS[curve], _p(_words(tau)), n, _p(out)) == 0
    return out


def commit(curve, ck_xy, coeffs, ck_inf=None):
    """ck_xy: uint64[n_ck, 2L]; coeffs: uint64[n,4].  Returns (xy[2L], inf) or raises ValueError."""
    L = lib().oracle_fp_limbs(CURVE_IDS[curve])
    out = np.zeros(2 * L, dtype=np.uint64)
    inf = np.zeros(1, dtype=np.uint8)
    rc = lib().oracle_commit(CURVE_IDS[curve], _p(ck_xy), _p(ck_inf), ck_xy.shape[0], _p(coeffs), coeffs.shape[0],
                             _p(out), _p(inf))
    if rc == -4:
        raise ValueError("Polynomial degree exceeds maximum allowed degree")
    assert rc == 0
    return out, int(inf[0])


def open_quotient(curve, polys, lens, z, xi):
    """polys: uint64[k, stride, 4].  Returns (quotient uint64[max_len-1, 4], eval int)."""
    k, stride = polys.shape[0], polys.shape[1]
    lens_a = np.asarray(lens, dtype=np.uint64)
    n = int(max(lens)) if len(lens) else 0
    quot = np.zeros((max(n - 1, 1), 4), dtype=np.uint64)
    ev = np.zeros(4, dtype=np.uint64)
    got = lib().oracle_open_quotient(CURVE_IDS[curve], _p(polys), _p(lens_a), k, stride, _p(_words(z)),
                                     _p(_words(xi)), _p(quot), _p(ev))
    assert got == n
    return quot[:max(n - 1, 0)], int.from_bytes(ev.tobytes(), "little")


def g1_mul(curve, xy, k, inf=False):
    L = lib().oracle_fp_limbs(CURVE_IDS[curve])
    out = np.zeros(2 * L, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    assert lib().oracle_g1_mul(CURVE_IDS[curve], _p(xy), int(inf), _p(_words(k)), _p(out), _p(oinf)) == 0
    return out, int(oinf[0])
