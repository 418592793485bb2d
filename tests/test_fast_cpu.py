"""oracle/fast_cpu.cpp (the multi-threaded CPU baseline bench.py quotes as cpu_baseline_optimised)
against oracle/py_oracle.py: it may only be timed beside the GPU once it computes the same things."""
import random

import numpy as np
import pytest

from oracle import fast_cpu as FC
from oracle import py_oracle as O


def limbs(vals, k=4):
    return np.frombuffer(b"".join(int(v).to_bytes(8 * k, "little") for v in vals), dtype="<u8").reshape(len(vals), k).copy()


def ints(a):
    a = np.ascontiguousarray(a, dtype="<u8")
    nb = 8 * a.shape[-1]
    raw = a.tobytes()
    return [int.from_bytes(raw[i:i + nb], "little") for i in range(0, len(raw), nb)]


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_ntt_matches_the_reference_recursion(curve):
    cv = O.curve(curve)
    rng = random.Random(3)
    for log_n in (0, 1, 2, 5, 10):
        n = 1 << log_n
        x = [rng.randrange(cv.r) for _ in range(n)]
        for w in (cv.root_of_unity(n), rng.randrange(2, cv.r)):           # primitive and arbitrary
            assert ints(FC.ntt(curve, limbs(x), w, threads=3)) == O.fft_ff(list(x), w, cv.r)
            assert ints(FC.ntt(curve, limbs(x), w, inverse=True, threads=2)) == O.ifft_ff(list(x), w, cv.r)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_setup_and_msm_match_the_oracle(curve):
    cv = O.curve(curve)
    L = 4 if curve == "bn254" else 6
    rng = random.Random(4)
    tau = rng.randrange(cv.r)
    n = 40
    ck = FC.setup(curve, tau, n, threads=4)
    ref = O.setup(n - 1, tau, cv)
    assert [tuple(ints(row.reshape(2, L))) for row in ck] == [O.normalize(p, cv) for p in ref]
    cases = [[rng.randrange(cv.r) for _ in range(n)],
             [0] * n,                                              # zero polynomial -> infinity (kzg.py:109)
             [1] * n, [cv.r - 1] * n,                              # largest digits / negative wrap
             [0, 5] + [0] * (n - 2),
             [(1 << 15) + (1 << 31) + (1 << 250)] * 7 + [0x8000] * (n - 7)]   # digits at the signed boundary
    for sc in cases:
        xy, inf = FC.msm(curve, ck, limbs(sc), threads=3)
        want = O.normalize(O.commit(ref, [sc], cv)[0], cv)
        assert (None if inf else tuple(ints(xy.reshape(2, L)))) == want
    # duplicate points, P + (-P), infinity flags
    dup = np.ascontiguousarray(np.tile(ck[3], (6, 1)))
    sc = [5, cv.r - 5, 7, 7, 1, 0]
    xy, inf = FC.msm(curve, dup, limbs(sc), threads=2)
    want = O.normalize(O.multiply(ref[3], 15, cv), cv)
    assert not inf and tuple(ints(xy.reshape(2, L))) == want
    flags = np.array([0, 0, 1, 0, 0, 0], dtype=np.uint8)
    xy, inf = FC.msm(curve, dup, limbs(sc), inf=flags, threads=2)
    assert tuple(ints(xy.reshape(2, L))) == O.normalize(O.multiply(ref[3], 8, cv), cv)


def test_msm_trapdoor_at_2p12():
    cv = O.BLS12_381
    n = 1 << 12
    rng = random.Random(12)
    tau = rng.randrange(cv.r)
    sc = [rng.randrange(cv.r) for _ in range(n)]
    ck = FC.setup("bls12_381", tau, n, threads=FC.max_threads())
    xy, inf = FC.msm("bls12_381", ck, limbs(sc), threads=FC.max_threads())
    assert not inf and tuple(ints(xy.reshape(2, 6))) == O.normalize(O.commit_trapdoor(sc, tau, cv), cv)
