#!/usr/bin/env python3
"""BASELINE config 4 (degree-2^24 commit + batched open, sharded by coefficient range) as far as
ONE GPU can show it:

  whole   the 2^24-coefficient job on a single MI355X (the n_gpus = 1 point of the config):
          2^24-point key (27 GiB table), commit and open, timed; trapdoor identity checked.
  rank    what ONE rank of a G-way range-sharded job does: commit of its 2^24/G coefficients
          against its key shard, open_shard_begin / open_shard_finish; the
          exchange between the two phases is one field element per rank and G partial points at the
          end.  Timed on this GPU; NOT a measurement of the G-GPU job.

    python tools/config4_rank.py [--log-n 24] [--world 8] [--curve bls12_381]

Prints one JSON line."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from kzg_snark_amd import _native
    from kzg_snark_amd.kzg import KZG
    from kzg_snark_amd.sharding import range_of
    kzg = KZG(args.curve)
    r = kzg.curve_order
    ctx = _native.get_context(args.curve)
    L = ctx.fp_limbs
    n = 1 << args.log_n
    tau = 0x6b7a675f736e61726b7a675f736e6172 % r
    z, xi = 0x1111111111111111111111111111 % r, 0x2222222222222222222222 % r
    tw, zw, xw = _native.int_to_words(tau), _native.int_to_words(z), _native.int_to_words(xi)
    g = torch.Generator(device="cpu").manual_seed(24)
    host = torch.randint(0, 1 << 62, (n, 4), generator=g, dtype=torch.int64)
    host[:, 3] >>= 3
    dev = host.to("cuda:0")
    torch.cuda.synchronize()
    out = {"what": "BASELINE config 4 on one GPU", "curve": args.curve, "log_n": args.log_n}

    def pt(xy):
        v = _native.limbs_to_ints(np.asarray(xy).reshape(2, L))
        return (v[0], v[1])

    # Horner over the limbs on the host would take minutes at 2^24 in Python; evaluate p(tau) on the device.
    def eval_at(t, count, point):
        return ctx.poly_eval(count, t.data_ptr(), point)

    def g1_times(k):
        p3 = kzg._g1.normalize(kzg.multiply(kzg.G1, k))
        return (int(p3[0]), int(p3[1]))

    # ---- whole job on one GPU
    t0 = time.perf_counter()
    srs = ctx.srs_generate(tw, n)
    ctx.synchronize()
    out["srs_setup_s"] = round(time.perf_counter() - t0, 3)
    ctx.commit_device(srs, dev.data_ptr(), [n], n)                      # warm (buffers, domains)
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        xy, inf = ctx.commit_device(srs, dev.data_ptr(), [n], n)
        ts.append(time.perf_counter() - t0)
    out["whole_commit_ms"] = round(min(ts) * 1e3, 2)
    p_tau = eval_at(dev, n, tau)
    assert inf[0] == 0 and pt(xy) == g1_times(p_tau), "commit(ck, p) != p(tau) * G1"
    ctx.open(srs, dev.data_ptr(), [n], n, zw, xw, device=True)
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        oxy, oinf, ev = ctx.open(srs, dev.data_ptr(), [n], n, zw, xw, device=True)
        ts.append(time.perf_counter() - t0)
    out["whole_open_ms"] = round(min(ts) * 1e3, 2)
    p_z = eval_at(dev, n, z)
    assert _native.limbs_to_ints(ev.reshape(1, 4))[0] == xi * p_z % r
    q_tau = xi * (p_tau - p_z) * pow(tau - z, -1, r) % r               # ((P(tau) - P(z)) / (tau - z)), P = xi * p
    assert oinf[0] == 0 and pt(oxy) == g1_times(q_tau), "open != trapdoor witness"
    del srs

    # ---- one rank of a `world`-way range-sharded job (the last-but-one rank: it has a carry and a successor)
    G = args.world
    rank = max(G - 2, 0)
    lo, hi = range_of(rank, G, n)
    m = hi - lo
    sl = dev[lo:hi].contiguous()
    cshard = ctx.srs_generate(tw, m, start=lo)                          # commit shard: tau^lo .. tau^(hi-1)
    start = 0 if rank == 0 else lo - 1
    oshard = ctx.srs_generate(tw, hi - 1 - start, start=start)          # open shard (quotient slice)
    ctx.synchronize()
    ctx.commit_device(cshard, sl.data_ptr(), [m], m)
    carry = 0x3333333333333333 % r                                      # stands in for the exchanged element
    cw = _native.int_to_words(carry)
    ts_c, ts_o = [], []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        cxy, cinf = ctx.commit_device(cshard, sl.data_ptr(), [m], m)
        ts_c.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        ctx.open_shard_begin(sl.data_ptr(), [m], m, zw, xw)
        ctx.open_shard_finish(oshard, zw, cw, rank == 0)
        ts_o.append(time.perf_counter() - t0)
    s_tau = eval_at(sl, m, tau) * pow(tau, lo, r) % r                    # sum_i p_(lo+i) tau^(lo+i)
    assert cinf[0] == 0 and pt(cxy) == g1_times(s_tau), "shard commit != trapdoor share"
    out.update({"world": G, "rank_coeffs": m, "rank_commit_ms": round(min(ts_c) * 1e3, 2),
                "rank_open_ms": round(min(ts_o) * 1e3, 2),
                "note": "rank_* are one rank's local work on this GPU; the G-GPU job adds the exchange of one "
                        "field element and G points per polynomial (not measured here)"})
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
