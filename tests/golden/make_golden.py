#!/usr/bin/env python3
"""Generates tests/golden/*.json.

The reference (swusjask/kzg-snark) ships NO golden vectors and cannot run here
(SageMath and py_ecc are absent: SURVEY.md 8c), so these vectors are produced by
oracle/py_oracle.py -- the pure-Python restatement of fft_ff.py / kzg.py -- and
serve as regression anchors for the C oracle and the HIP path.  They are data
(inputs and expected outputs), not reference code.

plonk_instance_n16.json is different: its columns are the CONTENTS of the
reference's own fixture constraint-system/PLONK_ARITHMETIZATION_INSTANCE.pkl as
decoded in SURVEY.md section 4 (the pickle itself needs sage.* classes and is
never unpickled here); the expected INTT coefficients and commitments for those
columns are computed by the oracle.

    python tests/golden/make_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import py_oracle as O  # noqa: E402

H = hex


def aff(pt, cv):
    n = O.normalize(pt, cv)
    return None if n is None else [H(n[0]), H(n[1])]


def ntt_vectors():
    out = []
    rng = random.Random(0x6b7a67)
    for name in ("bn254", "bls12_381"):
        cv = O.curve(name)
        for log_n in (0, 1, 2, 3, 4, 6):
            n = 1 << log_n
            w = cv.root_of_unity(n)
            x = [rng.randrange(cv.r) for _ in range(n)]
            out.append({"curve": name, "log_n": log_n, "w": H(w), "input": [H(v) for v in x],
                        "fft": [H(v) for v in O.fft_ff(x, w, cv.r)],
                        "ifft": [H(v) for v in O.ifft_ff(x, w, cv.r)]})
        # a root that is NOT primitive: fft_ff.py never checks (recursion semantics)
        n, w = 8, rng.randrange(2, cv.r)
        x = [rng.randrange(cv.r) for _ in range(n)]
        out.append({"curve": name, "log_n": 3, "w": H(w), "input": [H(v) for v in x],
                    "fft": [H(v) for v in O.fft_ff(x, w, cv.r)], "ifft": [H(v) for v in O.ifft_ff(x, w, cv.r)],
                    "note": "non-primitive w"})
    return out


def kzg_vectors():
    out = []
    rng = random.Random(0x736e61)
    for name in ("bn254", "bls12_381"):
        cv = O.curve(name)
        tau = rng.randrange(cv.r)
        ck = O.setup(7, tau, cv)
        polys = [[rng.randrange(cv.r) for _ in range(8)], [rng.randrange(cv.r) for _ in range(5)],
                 [0, 0, 3, 0, cv.r - 1], [1] * 8, []]
        z, xi = rng.randrange(cv.r), rng.randrange(cv.r)
        proof, pz = O.open_(ck, polys[:3], z, xi, cv)
        out.append({
            "curve": name, "tau": H(tau),
            "ck": [aff(p, cv) for p in ck],
            "polys": [[H(c) for c in p] for p in polys],
            "commitments": [aff(c, cv) for c in O.commit(ck, polys, cv)],
            "open": {"polys": [0, 1, 2], "z": H(z), "xi": H(xi), "proof": aff(proof, cv), "combined_eval": H(pz)},
        })
    return out


def plonk_instance():
    # decoded contents of constraint-system/PLONK_ARITHMETIZATION_INSTANCE.pkl (BN254 Fr, n = 16)
    cv = O.BN254
    r = cv.r
    cols = {
        "qM": [0, 0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0],
        "qL": [1, 1, 1, 1, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0],
        "qR": [0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0],
        "qO": [0, 0, 0, 0, 0] + [r - 1] * 10 + [0],
        "qC": [0] * 14 + [1, 0],
        "a": [7, 11, 13, 17, 19, 7, 13, 30, 77, 647, 418609, 418616, 4604776, 4605346, 2979658862, 0],
        "b": [0, 0, 0, 0, 0, 11, 17, 19, 570, 647, 7, 11, 570, 647, 1, 0],
        "c": [7, 11, 13, 17, 19, 77, 30, 570, 647, 418609, 418616, 4604776, 4605346, 2979658862, 2979658863, 0],
    }
    perm = [5, 21, 6, 22, 23, 26, 34, 38, 37, 25, 41, 42, 43, 44, 45, 16, 17, 18, 19, 20, 31, 27, 35, 36, 28, 29,
            32, 33, 39, 40, 30, 47, 0, 1, 2, 3, 4, 8, 7, 24, 9, 10, 11, 12, 13, 14, 46, 15]
    # sanity: every row satisfies qM*a*b + qL*a + qR*b + qO*c + qC + PI = 0 with PI = -x on rows 0..4
    for i in range(16):
        pi = -cols["a"][i] if i < 5 else 0
        g = (cols["qM"][i] * cols["a"][i] * cols["b"][i] + cols["qL"][i] * cols["a"][i] + cols["qR"][i] * cols["b"][i]
             + cols["qO"][i] * cols["c"][i] + cols["qC"][i] + pi) % r
        assert g == 0, i
    g16 = cv.root_of_unity(16)
    tau = 0x1234567
    ck = O.setup(15, tau, cv)
    interp = {k: O.fft_ff_interpolation(v, g16, r) for k, v in cols.items()}
    return {
        "curve": "bn254", "modulus": str(r), "n": 16, "omega": H(g16), "tau": H(tau),
        "columns": {k: [H(x) for x in v] for k, v in cols.items()},
        "perm": perm,
        "interpolated": {k: [H(x) for x in v] for k, v in interp.items()},
        "commitments": {k: aff(O.commit(ck, [v], cv)[0], cv) for k, v in interp.items()},
    }


def config1_vectors():
    """BASELINE config 1 size (degree-2^10 polynomial: NTT + commit), both curves, plus an opening
    of two polynomials at that size.  The key is given by its secret (the engine generates it;
    five of its points are stored as spot checks)."""
    from oracle import c_oracle           # same algorithms in C: the 2^10-point key in seconds
    out = []
    rng = random.Random(0x636667310a)
    n = 1 << 10
    for name in ("bn254", "bls12_381"):
        cv = O.curve(name)
        w = cv.root_of_unity(n)
        x = [rng.randrange(cv.r) for _ in range(n)]
        y = [rng.randrange(cv.r) for _ in range(n - 3)]
        tau = rng.randrange(cv.r)
        z, xi = rng.randrange(cv.r), rng.randrange(cv.r)
        ck = O.setup(n - 1, tau, cv)
        coeffs = O.ifft_ff(x, w, cv.r)
        commits = O.commit(ck, [x, coeffs], cv)
        assert O.eq(commits[0], O.commit_trapdoor(x, tau, cv), cv)
        proof, pz = O.open_(ck, [x, y], z, xi, cv)
        assert O.eq(proof, O.open_trapdoor([x, y], z, xi, tau, cv), cv)
        # the C restatement (oracle/kzg_oracle.c) must give the same key and the same commitment
        import numpy as np
        L = 6 if name == "bls12_381" else 4
        limbs = lambda vals, k: np.frombuffer(b"".join(int(v).to_bytes(8 * k, "little") for v in vals),   # noqa: E731
                                              dtype="<u8").reshape(len(vals), k).copy()
        ck_xy = c_oracle.setup(name, tau, n)
        for i in (0, 1, 2, 511, 1023):
            assert [int.from_bytes(ck_xy[i, j * L:(j + 1) * L].tobytes(), "little") for j in (0, 1)] == \
                list(O.normalize(ck[i], cv)), "C oracle key differs from the Python oracle's"
        cxy, cinf = c_oracle.commit(name, ck_xy, limbs(x, 4))
        assert cinf == 0 and [int.from_bytes(cxy[j * L:(j + 1) * L].tobytes(), "little") for j in (0, 1)] == \
            list(O.normalize(commits[0], cv)), "C oracle commitment differs from the Python oracle's"
        out.append({"curve": name, "log_n": 10, "w": H(w), "input": [H(v) for v in x],
                    "fft": [H(v) for v in O.fft_ff(x, w, cv.r)], "ifft": [H(v) for v in coeffs],
                    "tau": H(tau), "ck_spot": {str(i): aff(ck[i], cv) for i in (0, 1, 2, 511, 1023)},
                    "commit_input": aff(commits[0], cv), "commit_ifft": aff(commits[1], cv),
                    "open": {"second_poly": [H(v) for v in y], "z": H(z), "xi": H(xi), "proof": aff(proof, cv),
                             "combined_eval": H(pz)}})
    return out


def ragged_fft_vectors():
    """fft_ff / ifft_ff on lengths that are NOT powers of two: what the reference recursion returns
    (fft_ff.py:15-37 never checks the length).  The `marlin` entries have the call shape of
    marlin/prover.py:439-449: a coefficient list shorter than the domain K (trailing zeros dropped
    by list(poly)) with the generator of K (|K| = 32 for the reference's R1CS fixture)."""
    out = []
    rng = random.Random(0x726167)
    for name in ("bn254", "bls12_381"):
        cv = O.curve(name)
        for n in (2, 3, 5, 6, 7, 12, 100, 257, 513):
            w = rng.randrange(2, cv.r)
            x = [rng.randrange(cv.r) for _ in range(n)]
            out.append({"curve": name, "n": n, "w": H(w), "input": [H(v) for v in x],
                        "fft": [H(v) for v in O.fft_ff(list(x), w, cv.r)],
                        "ifft": [H(v) for v in O.ifft_ff(list(x), w, cv.r)]})
        gK = cv.root_of_unity(32)
        for n in (20, 31):
            x = [rng.randrange(cv.r) for _ in range(n)]
            out.append({"curve": name, "n": n, "w": H(gK), "input": [H(v) for v in x], "note": "marlin",
                        "fft": [H(v) for v in O.fft_ff(list(x), gK, cv.r)],
                        "ifft": [H(v) for v in O.ifft_ff(list(x), gK, cv.r)]})
    return out


def plonk_proof():
    """The whole proof of the reference's 16-gate instance for a fixed tau and fixed blinders
    (oracle/plonk_oracle.py: plonk/prover.py:24-212 restated, transcript.py restated), with the
    challenges and the prover's polynomials z, t, t_lo, t_mid, t_hi, r.  k1 = 2, k2 = 3 are the
    coset shifts kzg_snark_amd/plonk.py:Domain picks (the reference samples them at random,
    plonk/encoder.py:72-97; any valid pair serves)."""
    from oracle import plonk_oracle as P
    cv = O.BN254
    inst = plonk_instance()
    col = {k: [int(v, 16) for v in vals] for k, vals in inst["columns"].items()}
    w_full = col["a"] + col["b"] + col["c"]
    circuit = (col["qM"], col["qL"], col["qR"], col["qO"], col["qC"], inst["perm"], w_full[:5], w_full[5:])   # main.py:79
    n, g, k1, k2 = 16, cv.root_of_unity(16), 2, 3
    assert all(pow(k, n, cv.r) != 1 for k in (k1, k2)) and pow(k1 * pow(k2, -1, cv.r), n, cv.r) != 1
    tau = 0x1234567
    blinders = [0x1111 * (i + 3) + (i << 200) for i in range(11)]
    proof, ch, polys = P.prove_round(circuit, n, g, k1, k2, tau, blinders, cv)
    pt = lambda p: [H(p[0]), H(p[1]), p[2]]       # noqa: E731
    return {"curve": "bn254", "n": n, "g": H(g), "k1": k1, "k2": k2, "tau": H(tau),
            "blinders": [H(b) for b in blinders],
            "challenges": {k: H(v) for k, v in ch.items()},
            "polynomials": {k: [H(c) for c in v] for k, v in polys.items()},
            "proof": {"commitments": {k: pt(v) for k, v in proof["commitments"].items()},
                      "evaluations": {k: H(v) for k, v in proof["evaluations"].items()},
                      "kzg_proofs": {k: pt(v) for k, v in proof["kzg_proofs"].items()}}}


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for fn, make in (("ntt_vectors.json", ntt_vectors), ("kzg_vectors.json", kzg_vectors),
                     ("plonk_instance_n16.json", plonk_instance), ("config1_vectors.json", config1_vectors),
                     ("ragged_fft_vectors.json", ragged_fft_vectors), ("plonk_proof_n16.json", plonk_proof)):
        if only and fn not in only:
            continue
        data = make()
        with open(os.path.join(HERE, fn), "w") as f:
            json.dump(data, f, indent=1)
        print("wrote", fn)
