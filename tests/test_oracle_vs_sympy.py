"""The oracle against a THIRD-PARTY implementation (sympy, importable in this image; not the reference and not written
by this build): independent evidence for oracle/py_oracle.py, NOT a reference pin -- the reference (SageMath + py_ecc)
cannot run here and ships no vectors, so DESIGN.md section 5 keeps saying "parity unpinned".

  * sympy.discrete.transforms.ntt / intt (number-theoretic transform modulo a prime, root g^((p-1)/n) with g the
    smallest primitive root)                 == O.fft_ff / O.ifft_ff (fft_ff.py:3-58) with w = g^((r-1)/n)
  * sympy.ntheory.elliptic_curve (affine short-Weierstrass group law over GF(p))
                                             == O.multiply / O.add / O.double (py_ecc's projective law, kzg.py:27-49)
  * sympy.polys.galoistools.gf_div by (X - z) == O.poly_divide_linear (kzg.py:154), gf_eval == O.poly_eval
Both curves.  CPU only."""
import random

import pytest

sympy = pytest.importorskip("sympy")
from sympy.discrete.transforms import intt, ntt  # noqa: E402
from sympy.ntheory.elliptic_curve import EllipticCurve  # noqa: E402
from sympy.polys.domains import ZZ  # noqa: E402
from sympy.polys.galoistools import gf_div, gf_eval  # noqa: E402

from oracle import py_oracle as O  # noqa: E402

CURVES = ["bls12_381", "bn254"]


@pytest.mark.parametrize("curve", CURVES)
def test_transforms_against_sympy(curve):
    cv = O.curve(curve)
    r = cv.r
    assert sympy.primitive_root(r) == cv.mult_gen            # the generator behind Sage's nth_root
    rng = random.Random(0x73796d)
    for log_n in range(1, 9):
        n = 1 << log_n
        w = cv.root_of_unity(n)
        x = [rng.randrange(r) for _ in range(n)]
        if log_n == 3:
            x = [r - 1, 0, 1, r - 2, 0, 0, r - 1, 2]
        assert [int(v) for v in ntt(list(x), r)] == O.fft_ff(list(x), w, r), n
        assert [int(v) for v in intt(list(x), r)] == O.ifft_ff(list(x), w, r), n


@pytest.mark.parametrize("curve", CURVES)
def test_group_law_against_sympy(curve):
    cv = O.curve(curve)
    E = EllipticCurve(0, cv.b, modulus=cv.p)
    G = E(*cv.g1)
    Gp = O.from_affine(cv.g1)

    def affine(P):
        return None if P.z == 0 else (int(P.x), int(P.y))

    rng = random.Random(0x6563)
    scalars = [1, 2, 3, 5, 255, 256, cv.r - 1, cv.r - 2, (1 << 200) + 12345] + [rng.randrange(cv.r) for _ in range(4)]
    pts = {}
    for s in scalars:
        pts[s] = s * G
        assert affine(pts[s]) == O.normalize(O.multiply(Gp, s, cv), cv), s
    assert O.normalize(O.multiply(Gp, cv.r, cv), cv) is None and affine(cv.r * G) is None      # r G = O
    # add: distinct points, P + P (doubling branch), P + (-P), O + P
    a, b = scalars[-1], scalars[-2]
    Pa, Pb = O.multiply(Gp, a, cv), O.multiply(Gp, b, cv)
    assert O.normalize(O.add(Pa, Pb, cv), cv) == affine(pts[a] + pts[b])
    assert O.normalize(O.add(Pa, Pa, cv), cv) == affine(pts[a] + pts[a]) == O.normalize(O.double(Pa, cv), cv)
    assert O.normalize(O.add(Pa, O.neg(Pa, cv), cv), cv) is None and affine(pts[a] + (-pts[a])) is None
    assert O.normalize(O.add(O.Z1(), Pb, cv), cv) == affine(pts[b])
    # a commitment: sum_i c_i (tau^i G) against sympy's scalar multiplication and addition
    tau, coeffs = 0x1234567, [3, 0, cv.r - 1, rng.randrange(cv.r)]
    ck = O.setup(len(coeffs) - 1, tau, cv)
    acc = None
    for i, c in enumerate(coeffs):
        if c:
            term = c * (pow(tau, i, cv.r) * G)
            acc = term if acc is None else acc + term
    assert O.normalize(O.commit(ck, [coeffs], cv)[0], cv) == affine(acc)


@pytest.mark.parametrize("curve", CURVES)
def test_division_by_a_linear_factor_against_sympy(curve):
    r = O.curve(curve).r
    rng = random.Random(0x646976)
    for n in (1, 2, 3, 9, 64, 130):
        c = [rng.randrange(r) for _ in range(n)]
        c[-1] = c[-1] or 1
        for z in (0, 1, r - 1, rng.randrange(r)):
            dense = [v for v in c[::-1]]                               # sympy: highest coefficient first
            q, rem = gf_div(dense, [1, (-z) % r], r, ZZ)
            want_q, want_ev = O.poly_divide_linear(list(c), z, r)
            assert [int(v) for v in q][::-1] == want_q, (n, z)
            ev = int(rem[0]) if rem else 0
            assert ev == want_ev == O.poly_eval(c, z, r) == int(gf_eval(dense, z, r, ZZ)), (n, z)
