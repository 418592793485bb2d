#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X KZG engine (contract: see DESIGN.md section 6).

One "step" = one pass of the hot path over one batch of B = 4 synthetic degree-2^20
polynomials on BLS12-381 (a prover round commits 3-4 polynomials per call:
plonk/prover.py:89,113,136), inputs already resident in HBM:

    B INTTs of 2^20 evaluations each (what fft_ff_interpolation does, fft_ff.py:60-85)
    followed by one KZG.commit call on the B coefficient vectors against a 2^20-point SRS
    (kzg.py:80-120).

`value` = commits/sec = polynomials committed per second over the whole job (all ranks).  N > 1: one process per GPU, every
rank commits its own polynomial against a replicated SRS -- the path shards by polynomial
with no data-path collective, so scaling is "weak".

Also reported on the same line: NTT Fr-elements/sec (the second half of BASELINE.json's
metric), a roofline object for the dominant kernel (msm_accumulate, timed with HIP events
on the library's stream inside the timed region), one for the NTT passes, and the CPU
baseline (the oracle's C restatement of the reference algorithms, 1 core, bounded sample).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 20] [--curve bls12_381]

`--mode range` is a second workload, not the headline line: ONE degree-2^log_n polynomial and its key
split by coefficient range over the ranks, commit + batched open per step (BASELINE config 4, strong
scaling), every result checked against the trapdoor identities; see run_range().
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # see kzg_snark_amd/_native.py: one HW queue per pipeline stream

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
R_BLS = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
R_BN = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def root_of_unity(r, gen, n):
    return pow(gen, (r - 1) // n, r)


def cpu_baseline(curve, log_n, r, omega):
    """oracle/kzg_oracle.c (reference algorithms, single thread) on a bounded sample."""
    from oracle import c_oracle as CO
    n = 1 << log_n
    rs = np.random.RandomState(123)
    data = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64)
    data[:, 3] >>= np.uint64(3)
    t0 = time.perf_counter()
    CO.fft(curve, data, omega, inverse=True)
    t_ntt = time.perf_counter() - t0
    sample = min(n, 1 << 13)
    ck = CO.setup(curve, 0x1234567, 4)                      # any valid points; cost is per coefficient
    ck = np.ascontiguousarray(np.tile(ck, (sample // 4, 1)))
    t0 = time.perf_counter()
    CO.commit(curve, ck, np.ascontiguousarray(data[:sample]))
    t_commit_sample = time.perf_counter() - t0
    t_commit_full = t_commit_sample * (n / sample)
    return {
        "value": 1.0 / (t_ntt + t_commit_full),
        "unit": "commits/s",
        "cores": 1,
        "kind": "port",
        "sample": (f"oracle/kzg_oracle.c, 1 thread: recursive INTT of 2^{log_n} measured in full ({t_ntt:.2f} s) + "
                   f"naive double-and-add commit on a {sample}-coefficient slice ({t_commit_sample:.2f} s) scaled "
                   f"linearly x{n // sample} to 2^{log_n} coefficients"),
        "ntt_elements_per_s": n / t_ntt,
        "host_cpus": os.cpu_count(),
    }


def run_range(args, ctx, dev, world, rank, dist, torch, _native):
    """BASELINE config 4: one polynomial of 2^log_n coefficients and its commitment key split by
    contiguous coefficient range over the ranks.  Step = commit + open of that
    polynomial: every rank runs a whole local MSM per operation; what crosses ranks is one field
    element (open) and one point per rank per operation, added on the host by every rank."""
    from kzg_snark_amd.kzg import KZG
    from kzg_snark_amd.sharding import DistributedCommitter, range_of
    kzg = KZG(args.curve)
    r = kzg.curve_order
    L = ctx.fp_limbs
    n = 1 << args.log_n
    lo, hi = range_of(rank, world, n)
    m = hi - lo
    tau = 0x6b7a675f736e61726b7a675f736e6172 % r
    z, xi = 0x1111111111111111111111111111 % r, 0x2222222222222222222222 % r
    tw, zw, xw = _native.int_to_words(tau), _native.int_to_words(z), _native.int_to_words(xi)
    t0 = time.perf_counter()
    cshard = ctx.srs_generate(tw, m, start=lo)                      # tau^lo .. tau^(hi-1)
    start = 0 if rank == 0 else lo - 1
    oshard = ctx.srs_generate(tw, hi - 1 - start, start=start)      # key slice of the quotient's coefficients
    ctx.synchronize()
    t_srs = time.perf_counter() - t0
    g = torch.Generator(device="cpu").manual_seed(0x6b7a + rank)
    host = torch.randint(0, 1 << 62, (m, 4), generator=g, dtype=torch.int64)
    host[:, 3] >>= 3
    sl = host.to(dev)

    def to_pt(xy, inf):
        if inf:
            return kzg.Z1
        v = _native.limbs_to_ints(np.asarray(xy).reshape(2, L))
        return (v[0], v[1], 1)

    def commit_fn(_polys):
        xy, inf = ctx.commit_device(cshard, sl.data_ptr(), [m], m)
        return [to_pt(xy[0], inf[0])]

    def begin_fn():
        return _native.limbs_to_ints(ctx.open_shard_begin(sl.data_ptr(), [m], m, zw, xw).reshape(1, 4))[0]

    def finish_fn(carry, first):
        xy, inf, ev = ctx.open_shard_finish(oshard, zw, _native.int_to_words(carry), first)
        return to_pt(xy, inf[0]), (_native.limbs_to_ints(ev.reshape(1, 4))[0] if first else None)

    dc = DistributedCommitter(commit_fn, kzg.add, kzg.Z1)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def step():
        return dc.commit_range(None), dc.open_range(begin_fn, finish_fn, z, r, n)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        commitment, (proof, ev) = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # trapdoor identities, evaluated shard-wise on the devices: p(tau) = sum_g tau^lo_g * slice_g(tau)
    part = ctx.poly_eval(m, sl.data_ptr(), tau) * pow(tau, lo, r) % r
    parts = [part]
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, part)
    p_tau = sum(parts) % r
    g1 = kzg._g1

    def affine(pt):
        q = g1.normalize(pt)
        return (int(q[0]), int(q[1]))

    ok_commit = affine(commitment) == affine(kzg.multiply(kzg.G1, p_tau))
    q_tau = (xi * p_tau - ev) * pow(tau - z, -1, r) % r               # (P(tau) - P(z)) / (tau - z), P = xi * p
    ok_open = affine(proof) == affine(kzg.multiply(kzg.G1, q_tau))
    if rank == 0:
        print(json.dumps({
            "metric": "KZG commit + batched open per second, one degree-2^%d polynomial sharded by coefficient range"
                      % args.log_n,
            "value": args.steps / elapsed, "unit": "commit+open/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "integer: 13x30-bit limbs in u32 (381-bit Fp), 9x29-bit limbs (255-bit Fr), 64-bit multiply-add",
            "data": "synthetic",
            "config": {"workload": f"degree-2^{args.log_n} commit + open, {args.curve}, key and polynomial split by "
                                   f"coefficient range over {world} rank(s)",
                       "log_n": args.log_n, "curve": args.curve, "coefficients_per_rank": m,
                       "exchange": "one field element per rank (open) and one G1 point per rank per operation, "
                                   "all-gathered and added on the host"},
            "verified": {"commit_trapdoor": bool(ok_commit), "open_trapdoor": bool(ok_open)},
            "srs_setup_s": t_srs}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if (ok_commit and ok_open) else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--batch", type=int, default=4, help="polynomials per step (one commit call)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-isolated", action="store_true", help="skip the one-commit-at-a-time accumulate timing")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo: rehearsal of the "
                                                      "multi-rank path with all ranks on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--mode", default="batch", choices=("batch", "range"),
                    help="batch (default, the headline metric): every rank commits its own polynomials; "
                         "range: ONE degree-2^log_n polynomial and the key sharded by coefficient range over "
                         "the ranks, commit + batched open per step (BASELINE config 4; strong scaling)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)              # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    from kzg_snark_amd import _native
    ctx = _native.Context(args.curve, device=local_rank)
    # a dedicated (non-null) torch stream shared with the library: torch copies and the
    # library's kernels are then ordered on one queue
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    if args.mode == "range":
        return run_range(args, ctx, dev, world, rank, dist, torch, _native)

    r, gen = (R_BLS, 7) if args.curve == "bls12_381" else (R_BN, 5)
    fp_bytes = 48 if args.curve == "bls12_381" else 32
    log_n = args.log_n
    n = 1 << log_n
    omega = root_of_unity(r, gen, n)
    w_words = _native.int_to_words(omega)

    # synthetic workload: SRS [tau^i G1] generated on the device; uniform Fr evaluations (seeded per rank)
    tau = 0x6b7a675f736e61726b7a675f736e6172 % r
    t0 = time.perf_counter()
    srs = ctx.srs_generate(_native.int_to_words(tau), n)
    ctx.synchronize()
    t_srs = time.perf_counter() - t0
    g = torch.Generator(device="cpu").manual_seed(0x6b7a + rank)
    B = args.batch
    host = torch.randint(0, 1 << 62, (B, n, 4), generator=g, dtype=torch.int64)
    host[:, :, 3] >>= 3                                    # < 2^253 < r
    evals = host.to(dev)
    # two work buffers: a step's coefficients stay untouched while its commits are in flight.  A step
    # transforms its buffer in place, so the "evaluations" of step i+2 are the coefficients of step i:
    # an INTT is a bijection on Fr^n, the scalars stay uniform, and no copy sits in the timed region.
    works = [evals.clone(), evals]
    lens = [n] * B
    fp_limbs = ctx.fp_limbs
    results = {}

    def step(i):
        work = works[i & 1]
        ctx.ntt_device(work.data_ptr(), log_n, w_words, True, B)
        out_xy = np.zeros((B, 2 * fp_limbs), dtype=np.uint64)
        out_inf = np.zeros(B, dtype=np.uint8)
        results[i] = (out_xy, out_inf)
        ctx.commit_device_async(srs, work.data_ptr(), lens, n, out_xy, out_inf)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    ctx.commit_flush()
    ctx.prof_enable(True)
    ctx.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    ctx.commit_flush()                                      # every result is on the host before the clock stops
    barrier()
    elapsed = time.perf_counter() - t0
    assert all(int(inf.sum()) == 0 and xy.any() for xy, inf in results.values())
    ctx.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # The NTT half of the metric, timed alone (inside the pipelined loop above its kernels share the
    # GPU with the previous polynomial's bucket reduction, which inflates their event times).
    spans_main = {name: ctx.prof_read(name) for name in
                  ("msm_partition1", "msm_partition2", "msm_order", "msm_accumulate", "msm_finalize", "msm_reduce")}
    ctx.prof_enable(True)
    ctx.prof_reset()
    ntt_iters = max(args.steps, 10)
    for i in range(ntt_iters):
        ctx.ntt_device(works[0].data_ptr(), log_n, w_words, bool(i & 1) ^ True, B)
    barrier()
    ntt_alone = ctx.prof_read("ntt_pass")
    # The accumulate kernel alone: in the pipelined loop it deliberately shares every SIMD with the
    # next polynomial's prep and the previous one's reduce stage, so its span there is the pipeline
    # period.  One commit at a time (flush after each) gives the kernel's own duration.
    ctx.prof_reset()
    for p in range(0 if args.no_isolated else B):
        xy1 = np.zeros((1, 2 * fp_limbs), dtype=np.uint64)
        inf1 = np.zeros(1, dtype=np.uint8)
        ctx.commit_device_async(srs, works[0].data_ptr() + p * n * 32, [n], n, xy1, inf1)
        ctx.commit_flush()
    barrier()
    acc_alone = ctx.prof_read("msm_accumulate")
    ctx.prof_enable(False)

    spans = dict(spans_main)
    spans["ntt_pass"] = (ntt_alone[0] * args.steps / ntt_iters, ntt_alone[1] * args.steps // ntt_iters)
    if rank == 0:
        acc_ms, acc_cnt = spans["msm_accumulate"]
        acc_avg_s = (acc_ms / max(acc_cnt, 1)) * 1e-3
        msm_bytes = n * (32 + 2 * fp_bytes)                 # scalars + affine points, per commit
        ntt_ms, ntt_cnt = ntt_alone
        # one launch covers the whole batch; two launches (passes) per transform above 2^12
        ntt_per_transform_s = (ntt_ms / max(ntt_cnt, 1)) * (2e-3 if log_n > 12 else 1e-3) / B
        ntt_bytes = 2 * n * 32                              # read + write every element once
        out = {
            "metric": "KZG G1 commits/sec (INTT 2^%d + commit 2^%d, BLS12-381)" % (log_n, log_n)
            if args.curve == "bls12_381" else "KZG G1 commits/sec (INTT + commit, BN254)",
            "value": world * args.steps * B / elapsed,
            "unit": "commits/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "integer: 13x30-bit limbs in u32 (381-bit Fp), 9x29-bit limbs (255-bit Fr), 64-bit multiply-add",
            "data": "synthetic",
            "config": {"workload": f"degree-2^{log_n} INTT + KZG commit, {args.curve}, 2^{log_n}-point SRS, "
                                   f"uniform Fr scalars, batch of {B} polynomials per GPU per step",
                       "log_n": log_n, "curve": args.curve, "batch": B, "window_bits": 20 if n >= (1 << 18) else 16,
                       "sharding": "independent polynomials per rank, replicated SRS"},
            "ntt_elements_per_s": n / ntt_per_transform_s if ntt_per_transform_s > 0 else None,
            "ntt_ms": ntt_per_transform_s * 1e3,
            "kernel_ms_per_commit": {k: (v[0] / (args.steps * B)) for k, v in spans.items()},
            "srs_setup_s": t_srs,
            "roofline": {
                "kernel": "msm_accumulate_kernel",
                "bound": "hbm",
                "achieved": msm_bytes / acc_avg_s / 1e9 if acc_avg_s > 0 else None,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": (msm_bytes / acc_avg_s / 1e9) / HBM_PEAK_GBPS if acc_avg_s > 0 else None,
                "traffic": None,
                "algorithmic_bytes_per_launch": msm_bytes,
                "avg_launch_ms": acc_avg_s * 1e3,
                "isolated": {   # same kernel with nothing else on the GPU (one commit at a time)
                    "avg_launch_ms": acc_alone[0] / max(acc_alone[1], 1),
                    "achieved": msm_bytes / (acc_alone[0] / max(acc_alone[1], 1) * 1e-3) / 1e9 if acc_alone[0] > 0 else None,
                    "frac": msm_bytes / (acc_alone[0] / max(acc_alone[1], 1) * 1e-3) / 1e9 / HBM_PEAK_GBPS
                    if acc_alone[0] > 0 else None,
                },
                "note": "in the timed loop the persistent accumulate kernel runs beside prep(p+1) and reduce(p-1); "
                        "its span there is the pipeline period",
            },
            "roofline_ntt": {
                "kernel": "ntt_pass_kernel (2 launches per transform)",
                "bound": "hbm",
                "achieved": ntt_bytes / ntt_per_transform_s / 1e9 if ntt_per_transform_s > 0 else None,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": (ntt_bytes / ntt_per_transform_s / 1e9) / HBM_PEAK_GBPS if ntt_per_transform_s > 0 else None,
                "traffic": None,
                "algorithmic_bytes_per_transform": ntt_bytes,
            },
        }
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):                   # HBM bytes per launch from rocprofv3 --pmc passes
            tr = json.load(open(traffic_file))
            out["roofline"]["traffic"] = tr.get("msm_accumulate_kernel")
            out["roofline_ntt"]["traffic"] = tr.get("ntt_pass_kernel_per_transform")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.curve, log_n, r, omega)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
