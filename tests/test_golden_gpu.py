"""The HIP path against FROZEN data: every vector under tests/golden/ is replayed through the C ABI
and compared with the stored expected values -- no live oracle call, so the engine is pinned to
committed bytes rather than to whatever oracle/ computes today (the oracle is checked against the
same files on CPU in tests/test_oracle_pins.py).

  ntt_vectors.json           fft_ff / ifft_ff, n = 1 .. 64, primitive and non-primitive w
  ragged_fft_vectors.json    fft_ff / ifft_ff on lengths that are not powers of two (the reference
                             recursion's result, fft_ff.py:15-37; marlin/prover.py:439 call shape)
  kzg_vectors.json           commit / open against an 8-point key given as a list of points
  config1_vectors.json       BASELINE config 1 size: 2^10 NTT + commit + open, both curves
  plonk_instance_n16.json    the reference's own 16-gate fixture: 8 interpolations + 8 commitments
  plonk_proof_n16.json       the whole PLONK proof of that instance for fixed tau and blinders
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def ints(xs):
    return [int(v, 16) for v in xs]


def pt_of(entry):
    """stored [x, y] or null -> facade point."""
    return (1, 1, 0) if entry is None else (int(entry[0], 16), int(entry[1], 16), 1)


@pytest.fixture(scope="module")
def kzgs():
    from kzg_snark_amd.kzg import KZG
    return {c: KZG(c) for c in ("bn254", "bls12_381")}


@pytest.mark.parametrize("fname", ["ntt_vectors.json", "ragged_fft_vectors.json", "config1_vectors.json"])
def test_transforms_match_the_frozen_vectors(native, kzgs, fname):
    from kzg_snark_amd.fft_ff import fft_ff, ifft_ff
    seen = 0
    for v in load(fname):
        F = kzgs[v["curve"]].Fq
        x, w = [F(t) for t in ints(v["input"])], F(int(v["w"], 16))
        assert [int(t) for t in fft_ff(list(x), w, F)] == ints(v["fft"]), (fname, v["curve"], len(x), "fft")
        assert [int(t) for t in ifft_ff(list(x), w, F)] == ints(v["ifft"]), (fname, v["curve"], len(x), "ifft")
        # the buffer fast path and the C ABI directly
        arr = native.ints_to_limbs(ints(v["input"]))
        assert native.limbs_to_ints(fft_ff(arr, w, F)) == ints(v["fft"])
        seen += 1
    assert seen >= 2


def test_ragged_device_entry_point(native):
    """kzg_fft_ff_any_device on a device buffer (the host entry point is covered above)."""
    import ctypes
    import torch
    for v in load("ragged_fft_vectors.json"):
        if v["n"] not in (7, 100, 513):
            continue
        ctx = native.get_context(v["curve"])
        ctx.bind_torch_stream()
        for inverse, key in ((0, "fft"), (1, "ifft")):
            t = torch.from_numpy(native.ints_to_limbs(ints(v["input"])).view(np.int64)).to("cuda:0")
            rc = native.lib().kzg_fft_ff_any_device(ctx._h, ctypes.c_void_p(t.data_ptr()), v["n"],
                                                    native._as_vp(native.int_to_words(int(v["w"], 16))), inverse)
            assert rc == 0
            ctx.synchronize()
            assert native.limbs_to_ints(t.cpu().numpy().view(np.uint64)) == ints(v[key]), (v["curve"], v["n"], key)


def test_commit_and_open_match_the_frozen_vectors(kzgs):
    for v in load("kzg_vectors.json"):
        kzg = kzgs[v["curve"]]
        ck = [pt_of(p) for p in v["ck"]]                       # a plain list of points, as the reference's callers hold
        polys = [ints(p) for p in v["polys"]]
        assert kzg.commit(ck, polys) == [pt_of(c) for c in v["commitments"]]
        o = v["open"]
        sel = [polys[i] for i in o["polys"]]
        assert kzg.open(ck, sel, int(o["z"], 16), int(o["xi"], 16)) == pt_of(o["proof"])
        # and with the key generated on the device from the stored secret
        ck2, _ = kzg.setup(len(ck) - 1, tau=int(v["tau"], 16))
        assert [ck2[i] for i in range(len(ck))] == ck
        assert kzg.commit(ck2, polys) == [pt_of(c) for c in v["commitments"]]


def test_config1_commit_and_open(native, kzgs):
    """BASELINE config 1: degree-2^10 polynomial, NTT + commit (+ an opening of two polynomials)."""
    for v in load("config1_vectors.json"):
        kzg = kzgs[v["curve"]]
        n = 1 << v["log_n"]
        ck, _ = kzg.setup(n - 1, tau=int(v["tau"], 16))
        for i, p in v["ck_spot"].items():
            assert ck[int(i)] == pt_of(p), ("key point", i)
        x, coeffs = ints(v["input"]), ints(v["ifft"])
        assert kzg.commit(ck, [x, coeffs]) == [pt_of(v["commit_input"]), pt_of(v["commit_ifft"])]
        o = v["open"]
        assert kzg.open(ck, [x, ints(o["second_poly"])], int(o["z"], 16), int(o["xi"], 16)) == pt_of(o["proof"])
        # P(z) through the C ABI's eval_out
        ctx = native.get_context(v["curve"])
        arr = np.zeros((2, n, 4), dtype=np.uint64)
        arr[0] = native.ints_to_limbs(x)
        arr[1, :n - 3] = native.ints_to_limbs(ints(o["second_poly"]))
        _, _, ev = ctx.open(ck.srs, arr, [n, n - 3], n, native.int_to_words(int(o["z"], 16)),
                            native.int_to_words(int(o["xi"], 16)))
        assert native.limbs_to_ints(ev.reshape(1, 4))[0] == int(o["combined_eval"], 16)


def test_reference_fixture_columns(kzgs):
    """constraint-system/PLONK_ARITHMETIZATION_INSTANCE.pkl's columns (decoded):
    fft_ff_interpolation and KZG.commit of each, as plonk/indexer.py:60-77 runs them."""
    from kzg_snark_amd.fft_ff import fft_ff_interpolation
    g = load("plonk_instance_n16.json")
    kzg = kzgs[g["curve"]]
    F = kzg.Fq
    assert int(g["modulus"]) == kzg.curve_order
    ck, _ = kzg.setup(g["n"] - 1, tau=int(g["tau"], 16))
    omega = F(int(g["omega"], 16))
    for name, col in g["columns"].items():
        poly = fft_ff_interpolation([F(v) for v in ints(col)], omega, F)
        assert [int(c) for c in poly.list()] == ints(g["interpolated"][name]), name
        assert kzg.commit(ck, [poly])[0] == pt_of(g["commitments"][name]), name


def _circuit(g16):
    col = {k: ints(v) for k, v in g16["columns"].items()}
    w_full = col["a"] + col["b"] + col["c"]
    return col["qM"], col["qL"], col["qR"], col["qO"], col["qC"], g16["perm"], w_full[:5], w_full[5:]


@pytest.mark.parametrize("which", ["host", "device"])
def test_plonk_proof_is_bit_identical_to_the_frozen_one(which):
    """Same tau, same blinders => the same 7 commitments, 6 evaluations and 2 opening proofs as
    tests/golden/plonk_proof_n16.json (oracle/plonk_oracle.prove_round), and the same z / t / r
    coefficient vectors.  Covers the transcript (challenges are derived from the commitments)."""
    from kzg_snark_amd import plonk, plonk_device
    gp = load("plonk_proof_n16.json")
    circuit = _circuit(load("plonk_instance_n16.json"))
    curve, tau, blinders = gp["curve"], int(gp["tau"], 16), ints(gp["blinders"])
    trace = {}
    if which == "host":
        idx, prv = plonk.Indexer(curve), plonk.Prover(curve)
        ipk, ivk = idx.preprocess(*circuit[:6], tau=tau)
        get = lambda p: [int(c) for c in p.list()]                                   # noqa: E731
    else:
        idx = plonk_device.DeviceIndexer(curve)
        ipk, ivk = idx.preprocess(*circuit[:6], tau=tau)
        prv = plonk_device.DeviceProver(curve, alg=idx.alg)

        def get(t):
            c = idx.alg.download(t)
            while c and c[-1] == 0:
                c.pop()
            return c
    assert (int(ipk["subgroups"]["k1"]), int(ipk["subgroups"]["k2"])) == (gp["k1"], gp["k2"])
    proof = prv.prove(ipk, circuit[6], circuit[7], blinders=blinders, trace=trace)
    for k, v in gp["challenges"].items():
        assert int(trace[k]) == int(v, 16), f"challenge {k}"
    for k, v in gp["polynomials"].items():
        assert get(trace[k]) == ints(v), f"polynomial {k}"
    want = gp["proof"]
    for k, v in want["commitments"].items():
        assert tuple(int(c) for c in proof["commitments"][k]) == (int(v[0], 16), int(v[1], 16), v[2]), k
    for k, v in want["evaluations"].items():
        assert int(proof["evaluations"][k]) == int(v, 16), k
    for k, v in want["kzg_proofs"].items():
        assert tuple(int(c) for c in proof["kzg_proofs"][k]) == (int(v[0], 16), int(v[1], 16), v[2]), k
    assert plonk.Verifier(curve).verify(ivk, circuit[6], proof)
