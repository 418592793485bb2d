#!/usr/bin/env python3
"""Where a warm 2^k-gate PLONK prover round spends its wall time (cProfile over DeviceProver.prove; the GPU
synchronisation points show up inside the calls that wait).   python tools/plonk_profile.py [--log-n 20]"""
import argparse
import cProfile
import os
import pstats
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--top", type=int, default=45)
    args = ap.parse_args()
    from kzg_snark_amd import _native, plonk, plonk_device
    from kzg_snark_amd.kzg import KZG
    n = 1 << args.log_n
    Fq = KZG(args.curve).Fq
    qM, qL, qR, qO, qC, perm, x, w = plonk.synthetic_circuit(n, Fq, seed=args.log_n)
    idx = plonk_device.DeviceIndexer(args.curve)
    ipk, ivk = idx.preprocess(qM, qL, qR, qO, qC, perm)
    prv = plonk_device.DeviceProver(args.curve, alg=idx.alg)
    w_limbs = _native.ints_to_limbs([int(v) for v in w])
    prv.prove(ipk, x, w_limbs)
    prv.prove(ipk, x, w_limbs)
    pr = cProfile.Profile()
    pr.enable()
    prv.prove(ipk, x, w_limbs)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(args.top)


if __name__ == "__main__":
    main()
