"""kzg_snark_amd/csrc/field.h and ec.h -- the SAME headers the gfx950 kernels
compile -- built for the host with g++ and checked against Python integers /
the oracle.  CPU only: catches arithmetic bugs before any GPU time is spent."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle import py_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM_DIR = os.path.join(HERE, "shim")
FIELDS = [(0, O.BN254.r, 8), (1, O.BN254.p, 8), (2, O.BLS12_381.r, 8), (3, O.BLS12_381.p, 12)]


@pytest.fixture(scope="module")
def shim():
    so = os.path.join(SHIM_DIR, "libfield_shim.so")
    src = os.path.join(SHIM_DIR, "field_shim.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", so], check=True)
    return ctypes.CDLL(so)


def words(x, n):
    return (ctypes.c_uint32 * n)(*[(x >> (32 * i)) & 0xffffffff for i in range(n)])


def val(w):
    return sum(int(v) << (32 * i) for i, v in enumerate(w))


@pytest.mark.parametrize("fid,p,nw", FIELDS)
def test_field_ops(shim, fid, p, nw):
    rng = random.Random(fid)
    ops = [(0, lambda a, b: a * b % p), (1, lambda a, b: (a + b) % p), (2, lambda a, b: (a - b) % p),
           (4, lambda a, b: (-a) % p), (5, lambda a, b: 2 * a % p),
           (6, lambda a, b: ((((a + b) * (a - b) + a) * b) + b) % p),
           (7, lambda a, b: a * a % p), (8, lambda a, b: (a * b + (a + b) * (b - a)) % p),
           (9, lambda a, b: (a - b) * (a - b) % p), (10, lambda a, b: (b - a) % p), (11, lambda a, b: (-a) % p),
           (12, lambda a, b: a % p), (13, lambda a, b: 1 if (a - b) % p == 0 else 0),
           (14, lambda a, b: (a + 2 * b) * (a - b) % p), (15, lambda a, b: (2 * a - b) ** 2 % p),
           (16, lambda a, b: ((a - b) * (2 * a - b) - a * b) % p), (17, lambda a, b: (16 * a + 15 * b) % p),
           (18, lambda a, b: (-a - b) % p), (19, lambda a, b: (a - b) % p),
           (20, lambda a, b: (a * b + (a + b) * (a - b) + b * b + a * a + (a - b) * b + (a + b) * a) % p),
           # dot<3> + dot<1> + dot<2> over (a*b, (a+b)(b-a), -b*a)
           (21, lambda a, b: ((a * b + (a + b) * (b - a) - b * a) + a * b + (a * b + (a + b) * (b - a))) % p)]
    for _ in range(1500):
        a = rng.choice([0, 1, 2, p - 1, p - 2, rng.randrange(p), rng.randrange(p), 1 << (p.bit_length() - 1)])
        b = rng.choice([0, 1, p - 1, rng.randrange(p), rng.randrange(p)])
        for which, f in ops:
            out = (ctypes.c_uint32 * nw)()
            shim.shim_field_op(fid, which, words(a, nw), words(b, nw), out)
            assert val(out) == f(a, b), (which, hex(a), hex(b))
    for _ in range(10):
        a = rng.randrange(1, p)
        out = (ctypes.c_uint32 * nw)()
        shim.shim_field_op(fid, 3, words(a, nw), words(0, nw), out)
        assert val(out) == pow(a, -1, p)
    assert shim.shim_is_zero(fid, words(0, nw)) == 1
    assert shim.shim_is_zero(fid, words(p, nw)) == 1
    assert shim.shim_is_zero(fid, words(5, nw)) == 0


@pytest.mark.parametrize("cid,cv,nw", [(0, O.BN254, 8), (1, O.BLS12_381, 12)], ids=["bn254", "bls12_381"])
def test_ec_ops(shim, cid, cv, nw):
    """XYZZ madd / add / dbl incl. every edge case, against the oracle's group law."""
    rng = random.Random(cid + 40)
    g = O.from_affine(cv.g1)

    def pt_words(pt):
        n = O.normalize(pt, cv)
        if n is None:
            return words(0, 2 * nw), 1
        return words(n[0] | (n[1] << (32 * nw)), 2 * nw), 0

    def check(op, p1, p2, want):
        w1, i1 = pt_words(p1)
        w2, i2 = pt_words(p2)
        out = (ctypes.c_uint32 * (2 * nw))()
        oinf = ctypes.c_int(0)
        assert shim.shim_ec_op(cid, op, w1, i1, w2, i2, out, ctypes.byref(oinf)) == 0
        n = O.normalize(want, cv)
        if n is None:
            assert oinf.value == 1, op
        else:
            assert oinf.value == 0, op
            v = val(out)
            assert (v & ((1 << (32 * nw)) - 1), v >> (32 * nw)) == n, op

    a = O.multiply(g, rng.randrange(cv.r), cv)
    b = O.multiply(g, rng.randrange(cv.r), cv)
    inf = O.Z1()
    for op in (0, 1):                     # 0: madd (XYZZ + affine), 1: add (XYZZ + XYZZ)
        check(op, a, b, O.add(a, b, cv))
        check(op, a, a, O.double(a, cv))                   # P + P falls through to doubling
        check(op, a, O.neg(a, cv), inf)                    # P + (-P) = O
        check(op, inf, b, b)
        if op == 1:
            check(op, a, inf, a)
            check(op, inf, inf, inf)
    check(2, a, a, O.double(a, cv))       # dbl
    check(2, inf, inf, inf)
    # the MSM inner loop (flag-tracked infinity, sign folded into the lazy difference): a + b - b - a = O, then + b + b + a
    check(4, a, b, O.add(O.double(b, cv), a, cv))
    check(4, a, O.neg(a, cv), O.add(O.double(O.neg(a, cv), cv), a, cv))     # a - a (O) + a (copy) ... then -a -a + a
    # 48 flag-tracked steps (lazy X range of madd_finite), then a full add and a doubling of the result
    t = a
    for i in range(48):
        q = a if i % 3 == 0 else b
        t = O.add(t, O.neg(q, cv) if i % 5 == 4 else q, cv)
    check(5, a, b, O.double(O.add(t, b, cv), cv))
    # a non-trivial Z on the accumulator side: ((a + b) + b) + a
    check(3, a, b, O.add(O.add(O.add(a, b, cv), b, cv), a, cv))
