"""PLONK round trip (kzg_snark_amd/plonk.py) -- the harness of BASELINE config 5 at fixture scale.

CPU test: the protocol logic (encoder, quotient, linearisation, verifier equation) with the
oracle standing in for the engine's commit/open (test-only subclass), on the reference's own
16-gate instance (tests/golden/plonk_instance_n16.json) and a synthetic circuit.
GPU test: the same round through the real engine; mirrors the reference's end-to-end self-test
(plonk/verifier.py:216-292): a valid proof verifies, a tampered evaluation is rejected."""
import json
import os

import pytest

from oracle import py_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "plonk_instance_n16.json")


def fixture_instance():
    g = json.load(open(GOLD))
    col = {k: [int(x, 16) for x in v] for k, v in g["columns"].items()}
    w_full = col["a"] + col["b"] + col["c"]
    return col["qM"], col["qL"], col["qR"], col["qO"], col["qC"], g["perm"], w_full[:5], w_full[5:]   # main.py:79


def oracle_backed(curve):
    """KZG whose setup/commit/open run on the oracle (CPU) -- test scaffolding only."""
    from kzg_snark_amd.kzg import KZG
    cv = O.curve(curve)

    class OracleKZG(KZG):
        def setup(self, max_degree, tau=None):
            tau = int(self.Fq.random_element()) if tau is None else int(tau) % cv.r
            ck = [(p[0], p[1], 1) for p in (O.normalize(q, cv) for q in O.setup(max_degree, tau, cv))]
            return ck, self.multiply(self.G2, tau)

        def _to(self, pt):
            a = O.normalize(pt, cv)
            return self.Z1 if a is None else (a[0], a[1], 1)

        def commit(self, ck, polynomials):
            key = [O.Z1() if p[2] == 0 else (p[0], p[1], 1) for p in ck]
            return [self._to(c) for c in O.commit(key, [self._coeffs(p) for p in polynomials], cv)]

        def open(self, ck, polynomials, z, xi):
            key = [O.Z1() if p[2] == 0 else (p[0], p[1], 1) for p in ck]
            pr, _ = O.open_(key, [self._coeffs(p) for p in polynomials], int(self.Fq(z)), int(self.Fq(xi)), cv)
            return self._to(pr)

    return OracleKZG(curve)


@pytest.fixture
def cpu_interpolation(monkeypatch):
    """fft_ff_interpolation on the oracle for the CPU tests (the product's runs on the GPU)."""
    from kzg_snark_amd import plonk
    from kzg_snark_amd.field import PolynomialRing

    def interp(values, g, F):
        return PolynomialRing(F, "X")(O.fft_ff_interpolation([int(v) for v in values], int(g), F.p))

    monkeypatch.setattr(plonk, "fft_ff_interpolation", interp)


def round_trip(make_kzg, circuit, curve):
    from kzg_snark_amd import plonk
    qM, qL, qR, qO, qC, perm, x, w = circuit
    idx, prv, ver = plonk.Indexer(curve), plonk.Prover(curve), plonk.Verifier(curve)
    idx.kzg = prv.kzg = ver.kzg = make_kzg(curve)
    ipk, ivk = idx.preprocess(qM, qL, qR, qO, qC, perm)
    assert len(ipk["ck"]) == ipk["subgroups"]["n"] + 6                      # main.py:85
    proof = prv.prove(ipk, x, w)
    assert set(proof) == {"commitments", "evaluations", "kzg_proofs"}      # plonk/prover.py:188-210
    assert set(proof["commitments"]) == {"a", "b", "c", "z", "t_lo", "t_mid", "t_hi"}
    assert ver.verify(ivk, x, proof)
    bad = dict(proof)
    bad["evaluations"] = dict(proof["evaluations"])
    bad["evaluations"]["a"] = proof["evaluations"]["a"] + 1                 # plonk/verifier.py:277-290
    assert not ver.verify(ivk, x, bad)
    wrong_x = [x[0] + 1] + list(x[1:])
    assert not ver.verify(ivk, wrong_x, proof)


def test_fixture_instance_cpu(cpu_interpolation):
    round_trip(oracle_backed, fixture_instance(), "bn254")


def test_synthetic_circuit_cpu_bls(cpu_interpolation):
    from kzg_snark_amd import plonk
    from kzg_snark_amd.field import GF
    circuit = plonk.synthetic_circuit(8, GF(O.BLS12_381.r), seed=3)
    round_trip(oracle_backed, circuit, "bls12_381")


def test_unsatisfied_witness_is_caught(cpu_interpolation):
    from kzg_snark_amd import plonk
    qM, qL, qR, qO, qC, perm, x, w = fixture_instance()
    w = list(w)
    w[20] += 1
    idx, prv = plonk.Indexer("bn254"), plonk.Prover("bn254")
    idx.kzg = prv.kzg = oracle_backed("bn254")
    ipk, _ = idx.preprocess(qM, qL, qR, qO, qC, perm)
    with pytest.raises(AssertionError):
        prv.prove(ipk, x, w)


@pytest.mark.gpu
@pytest.mark.parametrize("curve,gates", [("bn254", None), ("bls12_381", 32)])
def test_round_trip_on_the_engine(curve, gates):
    from kzg_snark_amd import plonk
    from kzg_snark_amd.kzg import KZG
    from kzg_snark_amd.field import GF
    circuit = fixture_instance() if gates is None else plonk.synthetic_circuit(gates, GF(O.curve(curve).r), seed=7)
    round_trip(lambda c: KZG(c), circuit, curve)


@pytest.mark.gpu
@pytest.mark.parametrize("curve,gates", [("bn254", None), ("bls12_381", 64), ("bls12_381", 4096)])
def test_device_prover_round_trip(curve, gates):
    """kzg_snark_amd/plonk_device.py: the same protocol with every polynomial resident in HBM
    (coset-NTT quotient, batch-inverse accumulator).  Its proofs must satisfy the host Verifier."""
    from kzg_snark_amd import plonk, plonk_device
    from kzg_snark_amd.field import GF
    Fq = GF(O.curve(curve).r)
    if gates is None:
        qM, qL, qR, qO, qC, perm, x, w = fixture_instance()
    else:
        qM, qL, qR, qO, qC, perm, x, w = plonk.synthetic_circuit(gates, Fq, seed=gates)
    idx = plonk_device.DeviceIndexer(curve)
    ipk, ivk = idx.preprocess(qM, qL, qR, qO, qC, perm)
    prv = plonk_device.DeviceProver(curve, alg=idx.alg)
    proof = prv.prove(ipk, x, w)
    ver = plonk.Verifier(curve)
    assert ver.verify(ivk, x, proof)
    bad = dict(proof)
    bad["evaluations"] = dict(proof["evaluations"])
    bad["evaluations"]["c"] = proof["evaluations"]["c"] + 1
    assert not ver.verify(ivk, x, bad)
    from kzg_snark_amd import _native
    proof_l = prv.prove(ipk, x, _native.ints_to_limbs([int(v) for v in w]))     # witness handed over as limbs
    assert ver.verify(ivk, x, proof_l)
    w2 = list(w)
    w2[len(w2) // 2] = int(w2[len(w2) // 2]) + 1
    with pytest.raises(AssertionError):
        prv.prove(ipk, x, w2)                      # unsatisfied witness: the quotient has a remainder
