"""kzg_g1_sum: the host-side sum of the ranks' partial points (include/kzg_mi355x.h), against the oracle's group law.
A host function of the library -- no GPU involved -- so it runs in the CPU suite."""
import random

import pytest

from oracle import py_oracle as O


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_sum_of_partial_points(curve):
    from kzg_snark_amd import _native
    cv = O.curve(curve)
    g = O.from_affine(cv.g1)
    rng = random.Random(3)
    ks = [rng.randrange(cv.r) for _ in range(9)]
    P = [(p[0], p[1], 1) for p in (O.normalize(O.multiply(g, k, cv), cv) for k in ks)]
    assert _native.g1_sum(curve, P)[:2] == O.normalize(O.multiply(g, sum(ks) % cv.r, cv), cv)
    # what the exchange can deliver: no point at all, infinity records, the same partial twice, a partial and its
    # negative (kzg.py:116's running add handles each through py_ecc's add)
    assert _native.g1_sum(curve, []) == (1, 1, 0)
    assert _native.g1_sum(curve, [(1, 1, 0), P[0], (1, 1, 0)]) == P[0]
    assert _native.g1_sum(curve, [P[0], P[0]])[:2] == O.normalize(O.multiply(g, 2 * ks[0] % cv.r, cv), cv)
    neg = (P[0][0], (-P[0][1]) % cv.p, 1)
    assert _native.g1_sum(curve, [P[0], neg]) == (1, 1, 0)
    assert _native.g1_sum(curve, [P[0], neg, P[1]]) == P[1]
    # refused: a point off the curve, a coordinate that is not reduced
    for bad in ((P[0][0], (P[0][1] + 1) % cv.p, 1), (cv.p, 1, 1), (P[0][0] + cv.p, P[0][1], 1)):
        with pytest.raises(_native.NativeError):
            _native.g1_sum(curve, [bad])
