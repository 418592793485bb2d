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


if __name__ == "__main__":
    for fn, data in (("ntt_vectors.json", ntt_vectors()), ("kzg_vectors.json", kzg_vectors()),
                     ("plonk_instance_n16.json", plonk_instance())):
        with open(os.path.join(HERE, fn), "w") as f:
            json.dump(data, f, indent=1)
        print("wrote", fn)
