"""Host-side logic of the facade (kzg_snark_amd/field.py, curve.py, fft_ff.py, kzg.py)
that needs no GPU: the Sage-free field/polynomial shim, the py_ecc-shaped point
helpers, and the reference's error behaviour at the boundary."""
import random

import pytest

from kzg_snark_amd import curve as C
from kzg_snark_amd.field import GF, PolynomialRing
from oracle import py_oracle as O


def test_field_shim():
    F = GF(O.BLS12_381.r)
    a, b = F(5), F(-3)
    assert int(a + b) == 2 and int(a - 7) == O.BLS12_381.r - 2 and int(a * b) == O.BLS12_381.r - 15
    assert a / b * b == a and a ** (-1) * a == F(1) and int(2 - a) == O.BLS12_381.r - 3
    assert F(1) == 1 and F(0) == 0 and not F(0) and int(F(a)) == 5
    g = F.root_of_unity(1 << 10)
    assert int(g) == O.BLS12_381.root_of_unity(1 << 10)
    assert g.multiplicative_order() == 1 << 10
    assert (g * g).multiplicative_order() == 1 << 9
    assert int(F.multiplicative_generator()) == 7 and int(GF(O.BN254.r).multiplicative_generator()) == 5
    assert 0 <= int(F.random_element()) < O.BLS12_381.r


def test_polynomial_shim():
    F = GF(O.BN254.r)
    R = PolynomialRing(F, "X")
    X = R.gen()
    p = R([1, 2, 3, 0, 0])
    assert p.degree() == 2 and p.list() == [F(1), F(2), F(3)]
    assert R(0).degree() == -1 and R([]).list() == []
    assert p(2) == F(17)
    q = (p - p(5)) // (X - 5)
    assert q * (X - 5) + p(5) == p
    assert (p * p) % p == R(0) and (p * p) / p == p
    assert 3 * p == p * 3 == p + p + p
    assert F(2) * p == p + p
    rng = random.Random(4)
    a = R([rng.randrange(O.BN254.r) for _ in range(9)])
    b = R([rng.randrange(O.BN254.r) for _ in range(4)])
    qq, rr = divmod(a, b)
    assert qq * b + rr == a and rr.degree() < b.degree()


@pytest.mark.parametrize("name", ["bn254", "bls12_381"])
def test_point_helpers_match_oracle(name):
    cv, ocv = C.CURVES[name], O.curve(name)
    G = C.g1_group(cv)
    g = (cv.g1[0], cv.g1[1], 1)
    rng = random.Random(8)
    for _ in range(4):
        a, b = rng.randrange(cv.r), rng.randrange(cv.r)
        pa = G.multiply(g, a)
        assert pa[:2] == O.normalize(O.multiply(O.from_affine(ocv.g1), a, ocv), ocv) and pa[2] == 1
        assert G.add(pa, G.multiply(g, b)) == G.multiply(g, (a + b) % cv.r)
    assert G.add(g, G.neg(g)) == G.Z == (1, 1, 0)
    assert G.add(g, g) == G.double(g) == G.multiply(g, 2)
    assert G.multiply(g, 0) == G.Z and G.multiply(G.Z, 5) == G.Z and G.add(G.Z, g) == g
    assert G.eq((g[0] * 4 % cv.p, g[1] * 4 % cv.p, 4), g)       # un-normalised input is accepted
    G2 = C.g2_group(cv)
    g2 = (cv.g2[0], cv.g2[1], (1, 0))
    assert C.on_curve_g2(g2, cv) and G2.is_inf(G2.multiply(g2, cv.r))


def test_kzg_constructor_and_attributes():
    from kzg_snark_amd.kzg import KZG
    with pytest.raises(ValueError, match="Unsupported curve type"):      # kzg.py:37
        KZG("secp256k1")
    k = KZG()                                                             # default bn254 (kzg.py:18)
    assert k.curve_type == "bn254" and k.curve_order == O.BN254.r
    assert k.G1 == (1, 2, 1) and k.Z1 == (1, 1, 0)
    for attr in ("G1", "G2", "Z1", "Z2", "multiply", "add", "neg", "pairing", "eq", "curve_order", "Fq", "R", "X"):
        assert hasattr(k, attr)                                           # kzg.py:40-54
    assert k.X == k.R([0, 1]) and k.Fq(3) + 4 == 7
    assert k.add(k.multiply(k.G1, 3), k.neg(k.G1)) == k.multiply(k.G1, 2)
    kb = KZG("bls12_381")
    assert kb.curve_order == O.BLS12_381.r and kb.G1[:2] == O.BLS12_381.g1


def test_fft_ff_boundary_behaviour_without_gpu():
    from kzg_snark_amd.fft_ff import fft_ff, ifft_ff, fft_ff_interpolation
    F = GF(O.BLS12_381.r)
    g = F.root_of_unity(8)
    one = [F(9)]
    assert fft_ff(one, g, F) is one                                       # fft_ff.py:16-17
    assert ifft_ff(one, g, F) == one
    with pytest.raises(AssertionError, match="power of 2"):               # fft_ff.py:74
        fft_ff_interpolation([F(1)] * 6, g, F)
    with pytest.raises(AssertionError, match="Order of g"):               # fft_ff.py:78
        fft_ff_interpolation([F(1)] * 16, g, F)
    from kzg_snark_amd._native import NativeUnavailable
    with pytest.raises(NativeUnavailable):                                # length 3 is legal (the reference never
        fft_ff([F(1)] * 3, g, F)                                          # checks it): it needs the engine, no fallback
    with pytest.raises(RecursionError):                                   # fft_ff.py:16-26 on an empty list
        fft_ff([], g, F)
    with pytest.raises(ValueError):
        fft_ff([1, 2], 3, GF(101))                                        # not a supported scalar field


@pytest.mark.parametrize("name", ["bn254", "bls12_381"])
def test_pairing_bilinear(name):
    from kzg_snark_amd.pairing import pairing
    cv = C.CURVES[name]
    G1, G2 = C.g1_group(cv), C.g2_group(cv)
    g1, g2 = (cv.g1[0], cv.g1[1], 1), (cv.g2[0], cv.g2[1], (1, 0))
    e = pairing(g2, g1, cv)
    assert pairing(G2.multiply(g2, 6), G1.multiply(g1, 35), cv) == pairing(G2.multiply(g2, 210), g1, cv)
    assert pairing(g2, G1.multiply(g1, 210), cv) == pairing(G2.multiply(g2, 30), G1.multiply(g1, 7), cv)
    assert pairing(g2, G1.Z, cv) == pairing(G2.Z, g1, cv) != e          # e(O, Q) = e(P, O) = 1
    assert pairing(g2, G1.neg(g1), cv) != e


@pytest.mark.parametrize("name", ["bn254", "bls12_381"])
def test_check_and_batch_check_with_trapdoor_commitments(name):
    """KZG.check / batch_check (kzg.py:161-288) on commitments built from the trapdoor
    (commit(p) = p(tau) G1), so no GPU is needed; tampering must be rejected (kzg.py:361-380)."""
    from kzg_snark_amd.kzg import KZG
    kzg = KZG(name)
    Fq, r = kzg.Fq, kzg.curve_order
    rng = random.Random(17)
    tau = rng.randrange(r)
    rk = kzg.multiply(kzg.G2, tau)

    def ev(p, x):
        acc = 0
        for c in reversed(p):
            acc = (acc * x + c) % r
        return acc

    def instance(polys):
        z, xi = rng.randrange(r), rng.randrange(r)
        comms = [kzg.multiply(kzg.G1, ev(p, tau)) for p in polys]
        evals = [ev(p, z) for p in polys]
        comb_tau = sum(pow(xi, i + 1, r) * ev(p, tau) for i, p in enumerate(polys)) % r
        comb_z = sum(pow(xi, i + 1, r) * e for i, e in enumerate(evals)) % r
        proof = kzg.multiply(kzg.G1, (comb_tau - comb_z) * pow((tau - z) % r, -1, r) % r)
        return comms, z, evals, proof, xi

    insts = [instance([[rng.randrange(r) for _ in range(4)] for _ in range(2)]) for _ in range(3)]
    for comms, z, evals, proof, xi in insts:
        assert kzg.check(rk, comms, z, evals, proof, xi)
    args = [list(t) for t in zip(*insts)]
    assert kzg.batch_check(rk, *args)
    assert kzg.batch_check(rk, *args, r=Fq(12345))
    bad = [list(e) for e in args[2]]
    bad[0][0] = (bad[0][0] + 1) % r
    assert not kzg.check(rk, insts[0][0], insts[0][1], bad[0], insts[0][3], insts[0][4])
    assert not kzg.batch_check(rk, args[0], args[1], bad, args[3], args[4])


def test_transcript_bytes():
    import hashlib
    import struct
    from kzg_snark_amd.transcript import Transcript
    F = GF(O.BN254.r)
    t = Transcript("plonk-proof", F)
    s0 = hashlib.sha256(b"plonk-proof").digest()
    assert t.state == s0
    pt = (1, 2, 1)
    t.append_message("round1", [pt, 7, "x", F(5)])
    ser = str(pt).encode() + struct.pack(">q", 7) + b"x" + b"5"
    s1 = hashlib.sha256(s0 + b"round1" + ser).digest()
    assert t.state == s1
    c = t.get_challenge("beta")
    h = hashlib.sha256(s1 + b"beta").digest()
    assert int(c) == int.from_bytes(h, "big") % O.BN254.r
    assert t.state == hashlib.sha256(s1 + b"beta" + h).digest()
    with pytest.raises(struct.error):
        t.append_message("big", 1 << 70)          # the reference's ">q" packing has the same limit


def test_int_marshalling_helper_matches_the_python_forms():
    """csrc/pyconv.c (CPython-API int <-> limb conversion used by every list-taking facade call) against
    int.to_bytes / int.from_bytes, including the error cases int.to_bytes raises."""
    import random
    import numpy as np
    from kzg_snark_amd import _native, build
    from kzg_snark_amd.field import GF
    build.build_pyconv(verbose=False)
    pc = _native._load_pyconv()
    assert pc is not None, "the helper builds with the image's gcc + Python.h"
    rng = random.Random(11)
    for nb in (32, 48):
        vals = [0, 1, (1 << (8 * nb)) - 1, 1 << 63, (1 << 64) - 1, 1 << 64] + [rng.getrandbits(8 * nb) for _ in range(200)]
        raw = bytes(pc.ints_to_bytes(vals, nb))
        assert raw == b"".join(v.to_bytes(nb, "little") for v in vals)
        assert pc.bytes_to_ints(raw, nb) == vals
        assert pc.bytes_to_ints(np.frombuffer(raw, dtype=np.uint8), nb) == vals
    assert bytes(pc.ints_to_bytes([], 32)) == b"" and pc.bytes_to_ints(b"", 32) == []
    assert bytes(pc.ints_to_bytes((5, 6), 32)) == (5).to_bytes(32, "little") + (6).to_bytes(32, "little")
    F = GF(_native.R_BLS if hasattr(_native, "R_BLS") else 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001)
    assert pc.bytes_to_ints(bytes(pc.ints_to_bytes([F(7), F(-1)], 32)), 32) == [7, int(F(-1))]    # int()-able elements
    for bad in ([-1], [1 << 256]):
        with pytest.raises(OverflowError):
            pc.ints_to_bytes(bad, 32)
    with pytest.raises(TypeError):
        pc.ints_to_bytes([1.5j], 32)
    with pytest.raises(ValueError):
        pc.bytes_to_ints(b"123", 32)
    # and through the module-level functions the facade calls
    vals = [rng.getrandbits(255) for _ in range(1000)]
    limbs = _native.ints_to_limbs(vals)
    assert limbs.shape == (1000, 4) and limbs.dtype == np.uint64 and limbs.flags.writeable
    assert _native.limbs_to_ints(limbs) == vals
    assert _native.limbs_to_ints(limbs[::2]) == vals[::2]                                          # non-contiguous input
