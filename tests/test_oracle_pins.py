"""Pins the oracle (oracle/py_oracle.py, oracle/kzg_oracle.c) -- CPU only.

The reference holds no golden vectors for this path, so the
anchors are: the BN254 scalar modulus embedded in the reference's own
constraint-system fixtures, the published curve constants and known-answer
points, algebraic identities, and the committed tests/golden vectors."""
import json
import os
import random

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import py_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def limbs(vals, L=4):
    if not vals:
        return np.zeros((0, L), dtype=np.uint64)
    return np.frombuffer(b"".join(int(v).to_bytes(8 * L, "little") for v in vals), dtype="<u8").reshape(len(vals), L).copy()


def ints(a):
    a = np.ascontiguousarray(a)
    nb = 8 * a.shape[-1]
    raw = a.tobytes()
    return [int.from_bytes(raw[i:i + nb], "little") for i in range(0, len(raw), nb)]


def is_prime(n):
    if n < 2:
        return False
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def test_constants():
    # the only value-level anchor the reference ships: the modulus inside both .pkl fixtures
    assert load("plonk_instance_n16.json")["modulus"] == str(O.BN254.r)
    for cv in (O.BN254, O.BLS12_381):
        assert is_prime(cv.p) and is_prime(cv.r)
        assert O.is_on_curve(cv.g1, cv)
        assert O.is_inf(O.multiply(O.from_affine(cv.g1), cv.r, cv))
        assert (cv.r - 1) % (1 << cv.two_adicity) == 0 and (cv.r - 1) % (1 << (cv.two_adicity + 1)) != 0
        w = cv.root_of_unity(1 << 20)
        assert pow(w, 1 << 19, cv.r) == cv.r - 1
    assert O.BLS12_381.root_of_unity(1 << 20) == 0x3e1c54bcb947035a57a6e07cb98de4a2f69e02d265e09d9fece7e0e39898d4b
    assert O.BN254.root_of_unity(1 << 20) == 0x26125da10a0ed06327508aba06d1e303ac616632dbed349f53422da953337857
    # Published constants, independent of this repository: the 2-adic roots of unity the pairing libraries ship for
    # these scalar fields (arkworks / bellman `TWO_ADIC_ROOT_OF_UNITY`: generator^((r-1)/2^s) with the smallest
    # primitive root 7 resp. 5 -- the element Sage's Fq(1).nth_root(2^s) is recalled to return).
    # Every domain generator of the facade is a power of these.
    assert O.BLS12_381.root_of_unity(1 << 32) == \
        10238227357739495823651030575849232062558860180284477541189508159991286009131
    assert O.BN254.root_of_unity(1 << 28) == \
        19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert O.BLS12_381.root_of_unity(1 << 20) == pow(O.BLS12_381.root_of_unity(1 << 32), 1 << 12, O.BLS12_381.r)
    assert O.BN254.root_of_unity(1 << 20) == pow(O.BN254.root_of_unity(1 << 28), 1 << 8, O.BN254.r)


def test_known_answer_points():
    # 2*G1: EIP-196 test value for alt_bn128; BLS12-381 value from SURVEY.md 8c item 5
    two_g = O.normalize(O.double(O.from_affine(O.BN254.g1), O.BN254), O.BN254)
    assert two_g == (0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3,
                     0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)
    two_g = O.normalize(O.multiply(O.from_affine(O.BLS12_381.g1), 2, O.BLS12_381), O.BLS12_381)
    assert two_g == (
        0x572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e,
        0x166a9d8cabc673a322fda673779d8e3822ba3ecb8670e461f73bb9021d5fd76a4c56d9d4cd16bd1bba86881979749d28)


@pytest.mark.parametrize("cv", [O.BN254, O.BLS12_381], ids=["bn254", "bls12_381"])
def test_fft_is_the_dft_for_primitive_roots(cv):
    rng = random.Random(1)
    for n in (1, 2, 4, 8, 32, 128):
        w = cv.root_of_unity(n)
        x = [rng.randrange(cv.r) for _ in range(n)]
        assert O.fft_ff(x, w, cv.r) == O.dft_naive(x, w, cv.r)
        assert O.ifft_ff(O.fft_ff(x, w, cv.r), w, cv.r) == x
    single = [5]
    assert O.fft_ff(single, 3, cv.r) is single          # fft_ff.py:16-17


def test_interpolation_behaviour():
    cv = O.BN254
    g = cv.root_of_unity(8)
    vals = [3, 1, 4, 1, 5, 9, 2, 6]
    co = O.fft_ff_interpolation(vals, g, cv.r)
    for i, v in enumerate(vals):
        assert O.poly_eval(co, pow(g, i, cv.r), cv.r) == v
    assert O.fft_ff_interpolation([7] * 8, g, cv.r) == [7]           # trailing zeros dropped (fft_ff.py:85)
    with pytest.raises(AssertionError):
        O.fft_ff_interpolation(vals[:6], g, cv.r)                    # fft_ff.py:74
    with pytest.raises(AssertionError):
        O.fft_ff_interpolation(vals, cv.root_of_unity(4), cv.r)      # fft_ff.py:78


@pytest.mark.parametrize("cv", [O.BN254, O.BLS12_381], ids=["bn254", "bls12_381"])
def test_commit_open_identities(cv):
    rng = random.Random(2)
    tau = rng.randrange(cv.r)
    ck = O.setup(9, tau, cv)
    p = [rng.randrange(cv.r) for _ in range(10)]
    q = [rng.randrange(cv.r) for _ in range(4)]
    c = O.commit(ck, [p, q, [0, 0], []], cv)
    assert O.eq(c[0], O.commit_trapdoor(p, tau, cv), cv)
    assert O.eq(c[1], O.commit_trapdoor(q, tau, cv), cv)
    assert O.is_inf(c[2]) and O.is_inf(c[3])                        # zero polynomial -> Z1 (kzg.py:109)
    with pytest.raises(ValueError):
        O.commit(ck, [[1] * 11], cv)                                 # kzg.py:103-106
    O.commit(ck, [[1] * 10 + [0, 0]], cv)                            # trailing zeros are not degree
    z, xi = rng.randrange(cv.r), rng.randrange(cv.r)
    proof, pz = O.open_(ck, [p, q], z, xi, cv)
    assert O.eq(proof, O.open_trapdoor([p, q], z, xi, tau, cv), cv)
    # first polynomial is scaled by xi^1, not xi^0 (kzg.py:148-150)
    assert pz == (xi * O.poly_eval(p, z, cv.r) + xi * xi * O.poly_eval(q, z, cv.r)) % cv.r


@pytest.mark.parametrize("fname", ["ntt_vectors.json", "ragged_fft_vectors.json", "config1_vectors.json"])
def test_golden_ntt_vectors_both_oracles(fname):
    for v in load(fname):
        cv = O.curve(v["curve"])
        x = [int(s, 16) for s in v["input"]]
        w = int(v["w"], 16)
        assert [hex(t) for t in O.fft_ff(x, w, cv.r)] == v["fft"]
        assert [hex(t) for t in O.ifft_ff(x, w, cv.r)] == v["ifft"]
        d = limbs(x)
        CO.fft(v["curve"], d, w)
        assert [hex(t) for t in ints(d)] == v["fft"]
        d = limbs(x)
        CO.fft(v["curve"], d, w, inverse=True)
        assert [hex(t) for t in ints(d)] == v["ifft"]


def test_golden_kzg_vectors_both_oracles():
    for v in load("kzg_vectors.json"):
        name = v["curve"]
        cv = O.curve(name)
        L = 4 if name == "bn254" else 6
        tau = int(v["tau"], 16)
        ck = O.setup(7, tau, cv)
        assert [[hex(c) for c in O.normalize(p, cv)] for p in ck] == v["ck"]
        ck_xy = CO.setup(name, tau, 8)
        assert [[hex(a), hex(b)] for a, b in (ints(row.reshape(2, L)) for row in ck_xy)] == v["ck"]
        polys = [[int(c, 16) for c in p] for p in v["polys"]]
        for p, want in zip(polys, v["commitments"]):
            got = O.normalize(O.commit(ck, [p], cv)[0], cv)
            assert (None if got is None else [hex(got[0]), hex(got[1])]) == want
            xy, inf = CO.commit(name, ck_xy, limbs(p))
            got_c = None if inf else [hex(t) for t in ints(xy.reshape(2, L))]
            assert got_c == want
        o = v["open"]
        sel = [polys[i] for i in o["polys"]]
        z, xi = int(o["z"], 16), int(o["xi"], 16)
        proof, pz = O.open_(ck, sel, z, xi, cv)
        assert [hex(c) for c in O.normalize(proof, cv)] == o["proof"] and hex(pz) == o["combined_eval"]
        stride = max(len(p) for p in sel)
        arr = np.zeros((len(sel), stride, 4), dtype=np.uint64)
        for i, p in enumerate(sel):
            arr[i, :len(p)] = limbs(p)
        quot, ev = CO.open_quotient(name, arr, [len(p) for p in sel], z, xi)
        assert hex(ev) == o["combined_eval"]
        xy, inf = CO.commit(name, ck_xy, quot)
        assert [hex(t) for t in ints(xy.reshape(2, L))] == o["proof"]


def test_golden_config1_commit_and_open():
    """BASELINE config 1 size: the stored commitments / opening against the C oracle's naive commit
    (the Python oracle produced them) and the trapdoor identities."""
    for v in load("config1_vectors.json"):
        name = v["curve"]
        cv = O.curve(name)
        L = 4 if name == "bn254" else 6
        n, tau = 1 << v["log_n"], int(v["tau"], 16)
        x, coeffs = [int(s, 16) for s in v["input"]], [int(s, 16) for s in v["ifft"]]
        ck_xy = CO.setup(name, tau, n)
        for i, p in v["ck_spot"].items():
            assert [hex(t) for t in ints(ck_xy[int(i)].reshape(2, L))] == p
        for poly, key in ((x, "commit_input"), (coeffs, "commit_ifft")):
            xy, inf = CO.commit(name, ck_xy, limbs(poly))
            assert not inf and [hex(t) for t in ints(xy.reshape(2, L))] == v[key]
            assert [hex(t) for t in O.normalize(O.commit_trapdoor(poly, tau, cv), cv)] == v[key]
        o = v["open"]
        y, z, xi = [int(s, 16) for s in o["second_poly"]], int(o["z"], 16), int(o["xi"], 16)
        assert [hex(t) for t in O.normalize(O.open_trapdoor([x, y], z, xi, tau, cv), cv)] == o["proof"]
        arr = np.zeros((2, n, 4), dtype=np.uint64)
        arr[0], arr[1, :len(y)] = limbs(x), limbs(y)
        quot, ev = CO.open_quotient(name, arr, [n, len(y)], z, xi)
        assert hex(ev) == o["combined_eval"]
        xy, inf = CO.commit(name, ck_xy, quot)
        assert [hex(t) for t in ints(xy.reshape(2, L))] == o["proof"]


def test_golden_plonk_proof_is_what_the_oracle_round_produces():
    """tests/golden/plonk_proof_n16.json against oracle/plonk_oracle.prove_round, and the frozen
    proof accepted by a verifier equation evaluated with the trapdoor (no pairing needed:
    e(C - v G1 + z W, G2) = e(W, tau G2)  <=>  c - v + z w = tau w on the discrete logs)."""
    from oracle import plonk_oracle as P
    gp, g16 = load("plonk_proof_n16.json"), load("plonk_instance_n16.json")
    cv = O.curve(gp["curve"])
    col = {k: [int(x, 16) for x in v] for k, v in g16["columns"].items()}
    w_full = col["a"] + col["b"] + col["c"]
    circuit = (col["qM"], col["qL"], col["qR"], col["qO"], col["qC"], g16["perm"], w_full[:5], w_full[5:])
    proof, ch, polys = P.prove_round(circuit, gp["n"], int(gp["g"], 16), gp["k1"], gp["k2"], int(gp["tau"], 16),
                                     [int(b, 16) for b in gp["blinders"]], cv)
    assert {k: hex(v) for k, v in ch.items()} == gp["challenges"]
    assert {k: [hex(c) for c in v] for k, v in polys.items()} == gp["polynomials"]
    assert {k: [hex(v[0]), hex(v[1]), v[2]] for k, v in proof["commitments"].items()} == gp["proof"]["commitments"]
    assert {k: hex(v) for k, v in proof["evaluations"].items()} == gp["proof"]["evaluations"]
    assert {k: [hex(v[0]), hex(v[1]), v[2]] for k, v in proof["kzg_proofs"].items()} == gp["proof"]["kzg_proofs"]
    # every commitment is the trapdoor commitment of its polynomial
    tau = int(gp["tau"], 16)
    for k in ("a", "b", "c", "z", "t_lo", "t_mid", "t_hi"):
        want = O.normalize(O.commit_trapdoor(polys[k], tau, cv), cv)
        assert proof["commitments"][k] == (want[0], want[1], 1), k


def test_golden_plonk_instance():
    """Columns decoded from the reference's own PLONK fixture: the gate
    equation holds on every row, INTT coefficients and commitments match the vectors."""
    g = load("plonk_instance_n16.json")
    cv = O.BN254
    r = cv.r
    col = {k: [int(x, 16) for x in v] for k, v in g["columns"].items()}
    for i in range(16):
        pi = -col["a"][i] if i < 5 else 0
        assert (col["qM"][i] * col["a"][i] * col["b"][i] + col["qL"][i] * col["a"][i] + col["qR"][i] * col["b"][i]
                + col["qO"][i] * col["c"][i] + col["qC"][i] + pi) % r == 0
    w = col["a"] + col["b"] + col["c"]
    for i, j in enumerate(g["perm"]):                # copy constraints: w[i] == w[perm[i]]
        assert w[i] == w[j]
    omega = int(g["omega"], 16)
    ck_xy = CO.setup("bn254", int(g["tau"], 16), 16)
    for k, v in col.items():
        want = [int(x, 16) for x in g["interpolated"][k]]
        assert O.fft_ff_interpolation(v, omega, r) == want
        d = limbs(v)
        CO.fft("bn254", d, omega, inverse=True)
        assert O.poly_normalize(ints(d)) == want
        xy, inf = CO.commit("bn254", ck_xy, limbs(want))
        assert [hex(t) for t in ints(xy.reshape(2, 4))] == g["commitments"][k]


def test_c_oracle_medium_sizes_against_python():
    rng = random.Random(9)
    for name in ("bn254", "bls12_381"):
        cv = O.curve(name)
        n = 1 << 10                                   # BASELINE config 1: degree-2^10 NTT + commit on CPU
        w = cv.root_of_unity(n)
        x = [rng.randrange(cv.r) for _ in range(n)]
        d = limbs(x)
        CO.fft(name, d, w)
        assert ints(d) == O.fft_ff(x, w, cv.r)
        tau = rng.randrange(cv.r)
        ck_xy = CO.setup(name, tau, 64)
        L = ck_xy.shape[1] // 2
        p = x[:64]
        xy, inf = CO.commit(name, ck_xy, limbs(p))
        assert tuple(ints(xy.reshape(2, L))) == O.normalize(O.commit_trapdoor(p, tau, cv), cv)
