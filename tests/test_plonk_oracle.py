"""The PLONK prover's deterministic polynomials against the oracle restatement of
plonk/prover.py:214-414 (oracle/plonk_oracle.py) -- BASELINE config 5, SURVEY.md section 8f N2.

Blinders are injected, challenges are read back from the prover's trace, and the coefficient
vectors a, b, c, z, t, t_lo, t_mid, t_hi, r, PI are compared element-wise: the device prover
(batch-inverse accumulator, coset-NTT quotient) is no longer accepted on the word of the
product's own verifier alone.

CPU: the restatement against the identities the reference asserts (plonk/prover.py:110, :171,
:354) on the reference's own 16-gate instance; the host prover against the restatement.
GPU: the device prover against the restatement at n = 16 (fixture) and n = 2^10, and the
config-5-sized round (2^20 gates, BLS12-381): verifier accepts, tampering is rejected, and the
accumulator / quotient satisfy the reference's own asserts at random points."""
import random

import pytest

from oracle import plonk_oracle as P
from oracle import py_oracle as O
from test_plonk import fixture_instance, oracle_backed, cpu_interpolation  # noqa: F401

BLINDERS = [0x1111 * (i + 3) + (i << 200) for i in range(11)]


def oracle_polys(curve, circuit, sub, blinders, ch):
    """Run the restated prover pieces with the given blinders and challenges."""
    r = O.curve(curve).r
    qM, qL, qR, qO, qC, perm, x, w = circuit
    n, g, k1, k2 = sub["n"], int(sub["g"]), int(sub["k1"]), int(sub["k2"])
    full = [int(v) % r for v in list(x) + list(w)]
    m = len(full) // 3
    cols = [full[i * m:(i + 1) * m] + [0] * (n - m) for i in range(3)]
    H = [pow(g, i, r) for i in range(n)]
    if len(perm) != 3 * n:                                   # padding rows map to themselves
        mm = len(qM)
        fullp = list(range(3 * n))
        for blk in range(3):
            for i in range(mm):
                j = perm[blk * mm + i]
                fullp[blk * n + i] = (j // mm) * n + (j % mm)
        perm = fullp
    label = H + [k1 * h % r for h in H] + [k2 * h % r for h in H]        # plonk/encoder.py:139-147
    sigma_star = [label[perm[i]] for i in range(3 * n)]

    def interp(vals):
        vals = [int(v) % r for v in vals] + [0] * (n - len(vals))
        return O.fft_ff_interpolation(vals, g, r)

    sel = {k: interp(v) for k, v in (("qM", qM), ("qL", qL), ("qR", qR), ("qO", qO), ("qC", qC))}
    S = [interp(sigma_star[i * n:(i + 1) * n]) for i in range(3)]
    b = [int(v) % r for v in blinders]
    a_p = P.wire_polynomial(cols[0], g, n, b[0], b[1], r)
    b_p = P.wire_polynomial(cols[1], g, n, b[2], b[3], r)
    c_p = P.wire_polynomial(cols[2], g, n, b[4], b[5], r)
    PI = P.public_input_poly(x, n, g, r)
    beta, gamma, alpha, zeta = (int(ch[k]) % r for k in ("beta", "gamma", "alpha", "zeta"))
    z_p = P.permutation_polynomial(cols[0], cols[1], cols[2], sigma_star, beta, gamma, g, k1, k2, n, H,
                                   b[6], b[7], b[8], r)
    t_p = P.quotient_polynomial(a_p, b_p, c_p, z_p, sel["qM"], sel["qL"], sel["qR"], sel["qO"], sel["qC"],
                                S[0], S[1], S[2], alpha, beta, gamma, PI, n, g, k1, k2, r)
    t_lo, t_mid, t_hi = P.split_quotient(t_p, n, b[9], b[10], r)
    ev = {"a": P.p_eval(a_p, zeta, r), "b": P.p_eval(b_p, zeta, r), "c": P.p_eval(c_p, zeta, r),
          "s_sigma1": P.p_eval(S[0], zeta, r), "s_sigma2": P.p_eval(S[1], zeta, r),
          "z_omega": P.p_eval(z_p, zeta * g % r, r)}                            # plonk/prover.py:147-152
    r_p = P.linearization_polynomial(ev["a"], ev["b"], ev["c"], ev["s_sigma1"], ev["s_sigma2"], ev["z_omega"],
                                     sel["qM"], sel["qL"], sel["qR"], sel["qO"], sel["qC"], S[2], z_p,
                                     t_lo, t_mid, t_hi, alpha, beta, gamma, zeta, PI, n, k1, k2, r)
    return {"a": a_p, "b": b_p, "c": c_p, "z": z_p, "PI": PI, "t": t_p, "t_lo": t_lo, "t_mid": t_mid,
            "t_hi": t_hi, "r": r_p, "evaluations": ev}


FIX_SUB = {"bn254": None}


def small_domain(curve, n):
    from kzg_snark_amd.kzg import KZG
    from kzg_snark_amd.plonk import Domain
    d = Domain(KZG(curve).Fq, n)
    return {"n": d.n, "g": d.g, "k1": d.k1, "k2": d.k2}


def test_polynomial_product_matches_the_definition():
    r = O.BLS12_381.r
    rng = random.Random(5)
    for la, lb in ((1, 1), (3, 7), (40, 33), (1, 20)):
        a = [rng.randrange(r) for _ in range(la)]
        b = [rng.randrange(r) for _ in range(lb)]
        assert P.p_mul(a, b, r) == P.p_mul_schoolbook(a, b, r)
    a = [r - 1] * 50                                        # largest column sums
    assert P.p_mul(a, a, r) == P.p_mul_schoolbook(a, a, r)
    q, rem = P.p_divmod_vanishing(P.p_add(P.p_mul(a, P.vanishing(8, r), r), [5, 6], r), 8, r)
    assert q == P.p_norm(a) and rem == [5, 6]


def test_restatement_satisfies_the_references_own_asserts():
    """plonk/prover.py:110 (L1 (z - 1) = 0 mod v_H), :354 (split), :171 (r(zeta) = 0) on the
    reference's 16-gate instance with arbitrary blinders and challenges."""
    curve = "bn254"
    r = O.curve(curve).r
    rng = random.Random(16)
    sub = small_domain(curve, 16)
    ch = {k: rng.randrange(r) for k in ("beta", "gamma", "alpha", "zeta")}
    got = oracle_polys(curve, fixture_instance(), sub, BLINDERS, ch)
    n = 16
    L1 = P.first_lagrange(n, r)
    _, rem = P.p_divmod_vanishing(P.p_mul(L1, P.p_sub(got["z"], [1], r), r), n, r)
    assert rem == []                                                                  # :110
    assert P.p_eval(got["r"], ch["zeta"], r) == 0                                     # :171
    assert len(got["t"]) <= 3 * n + 6 and len(got["z"]) == n + 3 and len(got["a"]) == n + 2
    # an unsatisfied witness is not divisible (the reference's R(...) coercion at :297 would raise)
    qM, qL, qR, qO, qC, perm, x, w = fixture_instance()
    w = list(w)
    w[20] += 1
    with pytest.raises(ArithmeticError):
        oracle_polys(curve, (qM, qL, qR, qO, qC, perm, x, w), sub, BLINDERS, ch)


def as_list(p, r):
    return O.poly_normalize([int(c) % r for c in (p.list() if hasattr(p, "list") else p)])


def compare(trace, want, r, get=as_list):
    for k in ("a", "b", "c", "PI", "z", "t", "t_lo", "t_mid", "t_hi", "r"):
        assert get(trace[k], r) == want[k], f"polynomial {k} differs from the oracle restatement"
    for k, v in want["evaluations"].items():
        assert int(trace["evaluations"][k]) % r == v, f"evaluation {k}"


@pytest.mark.parametrize("curve,gates", [("bn254", None), ("bls12_381", 8)])
def test_host_prover_polynomials_match_the_oracle(cpu_interpolation, curve, gates):  # noqa: F811
    from kzg_snark_amd import plonk
    from kzg_snark_amd.field import GF
    r = O.curve(curve).r
    circuit = fixture_instance() if gates is None else plonk.synthetic_circuit(gates, GF(r), seed=11)
    idx, prv = plonk.Indexer(curve), plonk.Prover(curve)
    idx.kzg = prv.kzg = oracle_backed(curve)
    ipk, _ = idx.preprocess(*circuit[:6], tau=12345)
    trace = {}
    prv.prove(ipk, circuit[6], circuit[7], blinders=BLINDERS, trace=trace)
    want = oracle_polys(curve, circuit, ipk["subgroups"], BLINDERS, trace)
    compare(trace, want, r)


@pytest.mark.gpu
@pytest.mark.parametrize("curve,gates", [("bn254", None), ("bls12_381", 16), ("bls12_381", 1024), ("bn254", 1024)])
def test_device_prover_polynomials_match_the_oracle(curve, gates):
    """z (batch inversion + prefix product) and t (coset NTTs of size 4n, pointwise, division by
    Z_H) from kzg_snark_amd/plonk_device.py against plonk/prover.py:214-318 restated."""
    from kzg_snark_amd import plonk, plonk_device
    from kzg_snark_amd.field import GF
    r = O.curve(curve).r
    circuit = fixture_instance() if gates is None else plonk.synthetic_circuit(gates, GF(r), seed=gates + 1)
    idx = plonk_device.DeviceIndexer(curve)
    ipk, ivk = idx.preprocess(*circuit[:6], tau=987654321)
    prv = plonk_device.DeviceProver(curve, alg=idx.alg)
    trace = {}
    proof = prv.prove(ipk, circuit[6], circuit[7], blinders=BLINDERS, trace=trace)
    want = oracle_polys(curve, circuit, ipk["subgroups"], BLINDERS, trace)
    compare(trace, want, r, get=lambda t, r_: O.poly_normalize(idx.alg.download(t)))
    assert plonk.Verifier(curve).verify(ivk, circuit[6], proof)
    # commitments of those polynomials: trapdoor identity commit(ck, p) = p(tau) G1
    cv = O.curve(curve)
    for name in ("z", "t_lo", "t_mid", "t_hi"):
        pt = proof["commitments"][name]
        assert (int(pt[0]), int(pt[1])) == O.normalize(O.commit_trapdoor(want[name], 987654321, cv), cv), name


@pytest.mark.gpu
def test_config5_round_at_2p20_gates():
    """BASELINE config 5: the whole prover round at n = 2^20 gates on BLS12-381 with every
    polynomial resident in HBM (4 INTTs of size n, 13 + 1 NTTs of size 4n, 9 MSMs of n+2..n+6
    points, 2 openings).  The proof must satisfy the host verifier (2 pairings), a tampered
    evaluation and a wrong public input must be rejected (plonk/verifier.py:277-290), and the
    accumulator and quotient must satisfy the reference's own asserts (plonk/prover.py:110,
    :171, :354) -- checked at random points, where the full-size oracle would take hours."""
    from kzg_snark_amd import _native, plonk, plonk_device
    from kzg_snark_amd.kzg import KZG
    curve, log_n = "bls12_381", 20
    n = 1 << log_n
    kzg = KZG(curve)
    r = kzg.curve_order
    tau = 0x5eed5eed5eed5eed5eed5eed5eed5eed5eed % r
    qM, qL, qR, qO, qC, perm, x, w = plonk.synthetic_circuit(n, kzg.Fq, seed=log_n)
    idx = plonk_device.DeviceIndexer(curve)
    ipk, ivk = idx.preprocess(qM, qL, qR, qO, qC, perm, tau=tau)
    prv = plonk_device.DeviceProver(curve, alg=idx.alg)
    w_limbs = _native.ints_to_limbs([int(v) for v in w])
    trace = {}
    proof = prv.prove(ipk, x, w_limbs, blinders=BLINDERS, trace=trace)
    ver = plonk.Verifier(curve)
    assert ver.verify(ivk, x, proof)
    bad = dict(proof)
    bad["evaluations"] = dict(proof["evaluations"])
    bad["evaluations"]["a"] = proof["evaluations"]["a"] + 1
    assert not ver.verify(ivk, x, bad)
    assert not ver.verify(ivk, [x[0] + 1] + list(x[1:]), proof)
    bad = dict(proof)
    bad["commitments"] = dict(proof["commitments"], z=proof["commitments"]["t_lo"])
    assert not ver.verify(ivk, x, bad)

    alg = idx.alg
    ev = lambda name, pt: alg.eval(trace[name], pt % r)      # noqa: E731
    g, k1, k2 = int(ipk["subgroups"]["g"]), int(ipk["subgroups"]["k1"]), int(ipk["subgroups"]["k2"])
    beta, gamma, alpha, zeta = trace["beta"], trace["gamma"], trace["alpha"], trace["zeta"]
    # z(1) = 1, and z stays 1 after a full turn: the permutation product is 1 (plonk/prover.py:110 in point form)
    assert ev("z", 1) == 1
    # the split (plonk/prover.py:354): t(s) = t_lo(s) + s^n t_mid(s) + s^2n t_hi(s) at a random s
    rng = random.Random(20)
    s = rng.randrange(r)
    sn = pow(s, n, r)
    assert ev("t", s) == (ev("t_lo", s) + sn * ev("t_mid", s) + sn * sn % r * ev("t_hi", s)) % r
    # the quotient identity itself at s: numerator(s) == t(s) (s^n - 1), numerator per plonk/prover.py:297-313
    C = ipk["coeffs"]
    cev = lambda name: alg.eval(C[name], s)                  # noqa: E731
    a_s, b_s, c_s, z_s, zw_s = ev("a", s), ev("b", s), ev("c", s), ev("z", s), ev("z", s * g)
    gate = (a_s * b_s * cev("qM") + a_s * cev("qL") + b_s * cev("qR") + c_s * cev("qO") + ev("PI", s) + cev("qC")) % r
    p1 = (a_s + beta * s + gamma) * (b_s + beta * k1 * s + gamma) * (c_s + beta * k2 * s + gamma) % r * z_s % r
    p2 = ((a_s + beta * cev("S_sigma1") + gamma) * (b_s + beta * cev("S_sigma2") + gamma)
          * (c_s + beta * cev("S_sigma3") + gamma)) % r * zw_s % r
    L1s = (sn - 1) * pow(n * (s - 1) % r, -1, r) % r
    numer = (gate + alpha * (p1 - p2) + alpha * alpha % r * (z_s - 1) * L1s) % r
    assert numer == ev("t", s) * (sn - 1) % r
    assert ev("r", zeta) == 0                                                         # plonk/prover.py:171
    # the seven commitments against the trapdoor: commit(ck, p) = p(tau) G1
    for name in ("a", "b", "c", "z", "t_lo", "t_mid", "t_hi"):
        want = kzg._g1.normalize(kzg.multiply(kzg.G1, ev(name, tau)))
        pt = proof["commitments"][name]
        assert (int(pt[0]), int(pt[1])) == (int(want[0]), int(want[1])), name


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_verifier_public_input_evaluation_matches_the_interpolated_polynomial(curve):
    """Domain.public_input_at (what the verifier uses, O(len(x))) against the oracle's PI(X) = -sum x_i L_i(X)
    (plonk/encoder.py:237-257) evaluated at the same point, off and on the domain."""
    from kzg_snark_amd.kzg import KZG
    from kzg_snark_amd.plonk import Domain
    Fq = KZG(curve).Fq
    r = int(Fq.order()) if hasattr(Fq, "order") else KZG(curve).curve_order
    rng = random.Random(5)
    for n in (4, 16, 64):
        dom = Domain(Fq, n)
        assert dom._H is None                                    # nothing of size n is built for the verifier
        g = int(dom.g)
        x = [rng.randrange(r) for _ in range(min(5, n))]
        PI = P.public_input_poly(x, n, g, r)
        for zeta in (rng.randrange(r), rng.randrange(r), pow(g, 2, r), pow(g, n - 1, r), 1):
            want = sum(c * pow(zeta, i, r) for i, c in enumerate(PI)) % r
            assert int(dom.public_input_at(x, zeta)) == want, (curve, n, zeta)
        assert int(dom.public_input_at([], 12345)) == 0
