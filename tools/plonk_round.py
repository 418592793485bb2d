#!/usr/bin/env python3
"""BASELINE config 5: a full PLONK index -> prove -> verify round on a synthetic mul/add-chain
circuit of 2^k gates, every polynomial resident on the GPU (kzg_snark_amd/plonk_device.py).

    python tools/plonk_round.py [--log-n 20] [--curve bls12_381]

Prints one JSON line with the wall-clock split.  The circuit generator and the key material are
Python-side (seconds at 2^20); the prover round itself is what the engine accelerates."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--curve", default="bls12_381")
    args = ap.parse_args()
    from kzg_snark_amd import plonk, plonk_device
    from kzg_snark_amd.kzg import KZG
    n = 1 << args.log_n
    Fq = KZG(args.curve).Fq
    t = {}
    t0 = time.perf_counter()
    qM, qL, qR, qO, qC, perm, x, w = plonk.synthetic_circuit(n, Fq, seed=args.log_n)
    t["circuit_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx = plonk_device.DeviceIndexer(args.curve)
    ipk, ivk = idx.preprocess(qM, qL, qR, qO, qC, perm)
    t["index_s"] = time.perf_counter() - t0
    prv = plonk_device.DeviceProver(args.curve, alg=idx.alg)
    t0 = time.perf_counter()
    proof = prv.prove(ipk, x, w)                      # witness as a Python list: includes int -> limb marshalling
    t["prove_s"] = time.perf_counter() - t0
    from kzg_snark_amd import _native
    w_limbs = _native.ints_to_limbs([int(v) for v in w])     # what a native witness generator would hand over
    t0 = time.perf_counter()
    proof = prv.prove(ipk, x, w_limbs)                # second proof: domains and buffers warm, witness in limb form
    t["prove_warm_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ok = plonk.Verifier(args.curve).verify(ivk, x, proof)
    t["verify_s"] = time.perf_counter() - t0
    print(json.dumps({"what": "PLONK round, device-resident prover", "curve": args.curve, "gates": n,
                      "verified": bool(ok), **{k: round(v, 3) for k, v in t.items()}}), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
