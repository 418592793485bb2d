#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X KZG engine (contract: see DESIGN.md section 6).

One "step" = one pass of the hot path over one batch of B = 4 synthetic degree-2^20
polynomials on BLS12-381 (a prover round commits 3-4 polynomials per call:
plonk/prover.py:89,113,136), inputs already resident in HBM:

    B INTTs of 2^20 evaluations each (what fft_ff_interpolation does, fft_ff.py:60-85)
    followed by one KZG.commit call on the B coefficient vectors against a 2^20-point SRS
    (kzg.py:80-120).

`value` = commits/sec = polynomials committed per second over the whole job (all ranks).  N > 1: one
process per GPU, every rank commits its own polynomials against a replicated SRS -- the path shards by
polynomial with no data-path collective, so scaling is "weak".  The last step's
commitments are checked against the trapdoor identity commit(ck, p) = p(tau) G1 after the clock stops.

The same JSON line also carries
  * ntt_elements_per_s + roofline_ntt   the NTT half of BASELINE.json's metric, timed alone;
  * roofline                             the dominant kernel (msm_accumulate), HIP events on its stream;
  * open                                 KZG.open of k = 6 polynomials of 2^20 (plonk/prover.py:184): opens/s,
                                         roofline of the lincomb/evaluate/divide kernels, verified;
  * range_mode                           BASELINE config 4: ONE degree-2^24 polynomial set and its key split by
                                         coefficient range over the ranks, commit + batched open per step,
                                         verified against the trapdoor identities (n_gpus = 1: the whole job);
  * distributed_ntt                      (N >= 2) the four-step 2^24 transform over RCCL all-to-all, verified
                                         against the single-GPU transform;
  * range_mode_2p20 / distributed_ntt_2p20   (N >= 2) the same two at the metric's own degree 2^20: strong scaling of
                                         ONE commit + open and ONE transform over the ranks;
  * plonk_round                          (N = 1) BASELINE config 5: index -> prove -> verify of a synthetic 2^20-gate
                                         circuit, prover resident on the GPU; ms per prover round, verifier accepts;
  * cpu_baseline / cpu_baseline_optimised   rank 0 at N = 1: the oracle's C restatement of the reference
                                         algorithms on 1 core, and a multi-threaded Montgomery + Pippenger
                                         CPU implementation on all host cores.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 20] [--curve bls12_381]

--gpus N > 1 without a launcher (no WORLD_SIZE in the environment) starts N child processes, one per
GPU, before this process has touched HIP; under torch.distributed.run it uses the ranks it is given
and refuses to run when WORLD_SIZE != --gpus.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # see kzg_snark_amd/_native.py: one HW queue per pipeline stream
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SHADER_PEAK_MHZ = 2400.0
VALU_PEAK_GINSTR = 1024 * 2.4 / 4   # wave64 VALU instructions/ns: 1024 SIMDs x 2.4 GHz / 4 cycles per instruction
R_BLS = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
R_BN = 21888242871839275222246405745257275088548364400416034343698204186575808495617
TAU = 0x6b7a675f736e61726b7a675f736e6172
Z_PT = 0x1111111111111111111111111111
XI_PT = 0x2222222222222222222222
DTYPE = "integer: 13x30-bit limbs in u32 (381-bit Fp), 9x29-bit limbs (255-bit Fr), 64-bit multiply-add"


def root_of_unity(r, gen, n):
    return pow(gen, (r - 1) // n, r)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--batch", type=int, default=4, help="polynomials per step (one commit call)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-isolated", action="store_true", help="skip the one-commit-at-a-time accumulate timing")
    ap.add_argument("--no-open", action="store_true", help="skip the KZG.open section")
    ap.add_argument("--no-range", action="store_true", help="skip the range-sharded config-4 section")
    ap.add_argument("--no-dist-ntt", action="store_true", help="skip the distributed NTT section")
    ap.add_argument("--no-plonk", action="store_true", help="skip the config-5 PLONK round (one GPU only)")
    ap.add_argument("--plonk-log-n", type=int, default=20, help="gates of the PLONK round (config 5: 2^20)")
    ap.add_argument("--plonk-mode", default="auto", choices=("auto", "dealt", "sharded"),
                    help="N > 1: one proof over the ranks with its VECTOR work split as well (sharded; auto picks it "
                         "for power-of-two world sizes) or with only the MSMs dealt over replicated vectors (dealt)")
    ap.add_argument("--range-log-n", type=int, default=24, help="degree of the range-sharded job (config 4: 24)")
    ap.add_argument("--range-steps", type=int, default=5)
    ap.add_argument("--open-k", type=int, default=6, help="polynomials per opening (plonk/prover.py:184 opens 6)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo: rehearsal of the "
                                                      "multi-rank path with all ranks on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="one GPU: bring up a ONE-rank process group and issue every collective the N-GPU run issues "
                         "(RCCL call path, dtypes, stream ordering); times nothing worth quoting")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="launcher self-test without a GPU: the ranks rendezvous over gloo, all-gather one record "
                         "each and rank 0 prints one JSON line")
    ap.add_argument("--mode", default="all", choices=("all", "batch", "range"),
                    help="all (default): headline + every section; batch: headline only; range: the config-4 "
                         "section only, printed as its own line")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: one process per GPU, started before anything here touches HIP
# ------------------------------------------------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """torch.cuda.device_count() in a short-lived child: this process must stay HIP-free, its
    children are what touch the GPU (an exec/fork after HIP initialisation is not allowed here)."""
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        return 0


def launch(args, argv):
    n = args.gpus
    if not (args.one_device or args.rehearse_launch):
        have = visible_gpus()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible; refusing to run a smaller job "
                  f"under that label", file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in procs:          # a rank failed: do not leave the others in a collective
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    return rc


def rehearse_launch(args, world, rank):
    """No GPU: proves that the launcher's children rendezvous and that the fixed-size record
    exchange works over gloo (tests/test_bench_launcher.py)."""
    import torch.distributed as dist
    from kzg_snark_amd.sharding import all_gather_bytes
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    got = all_gather_bytes(bytes([rank]) * 4)
    ok = got == [bytes([g]) * 4 for g in range(world)]
    if rank == 0:
        print(json.dumps({"launcher": "ok" if ok else "exchange mismatch", "world": world, "n_gpus": args.gpus}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


# ------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only; bounded samples)
# ------------------------------------------------------------------------------------------------

def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(curve, log_n, omega):
    """oracle/kzg_oracle.c (reference algorithms, single thread) on a bounded sample."""
    import numpy as np
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
        "cpu_model": cpu_model(),
    }


def cpu_baseline_optimised(curve, log_n, omega):
    """oracle/fast_cpu.c: Montgomery limbs with unsigned __int128, iterative NTT, signed-window Pippenger,
    OpenMP over all host cores -- the 'fair' CPU figure of BASELINE.md section 3.  Whole 2^log_n job."""
    import numpy as np
    try:
        from oracle import fast_cpu as FC
    except Exception as e:  # noqa: BLE001
        return {"error": f"oracle/fast_cpu unavailable: {e!r}"}
    n = 1 << log_n
    # the threads this process may really use: its affinity mask, capped at a one-GPU box's CPU share
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    rs = np.random.RandomState(321)
    data = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64)
    data[:, 3] >>= np.uint64(3)
    t0 = time.perf_counter()
    pts = FC.setup(curve, 0x1234567, n, threads=cores)
    t_setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    FC.ntt(curve, data, omega, inverse=True, threads=cores)
    t_ntt = time.perf_counter() - t0
    t0 = time.perf_counter()
    FC.msm(curve, pts, data, threads=cores)
    t_msm = time.perf_counter() - t0
    return {
        "value": 1.0 / (t_ntt + t_msm),
        "unit": "commits/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"oracle/fast_cpu.cpp, {cores} threads (OpenMP): iterative INTT of 2^{log_n} ({t_ntt * 1e3:.0f} ms) + "
                   f"Pippenger MSM of 2^{log_n} points ({t_msm:.2f} s), both measured in full, one repetition; "
                   f"key generated in {t_setup:.1f} s"),
        "ntt_elements_per_s": n / t_ntt,
        "cpu_model": cpu_model(),
    }


# ------------------------------------------------------------------------------------------------
# sections
# ------------------------------------------------------------------------------------------------

class Env:
    """What every section needs: the context, the device, ranks, helpers."""

    def __init__(self, args, torch, dist, ctx, dev, world, rank, native):
        self.args, self.torch, self.dist, self.ctx, self.dev = args, torch, dist, ctx, dev
        self.world, self.rank, self.native = world, rank, native
        self.collective = world > 1 or (dist is not None and dist.is_initialized())   # one-rank rehearsal included
        from kzg_snark_amd.kzg import KZG
        self.kzg = KZG(args.curve)
        self.r = self.kzg.curve_order

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.collective:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, seconds):
        if not self.collective:
            return seconds
        t = self.torch.tensor([seconds], device=self.dev if self.args.backend == "nccl" else "cpu",
                              dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def collectives_info(self):
        """Backend, the world size torch.distributed reports, and the device every rank sits on (gathered)."""
        dist, torch = self.dist, self.torch
        if not self.collective:
            return {"backend": None, "world_size_seen": 1, "device_per_rank": [str(self.dev)]}
        mine = f"{self.dev}:{torch.cuda.get_device_properties(self.dev).name}".encode()[:63].ljust(64, b" ")
        from kzg_snark_amd.sharding import all_gather_bytes
        return {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                "device_per_rank": [b.decode().strip() for b in all_gather_bytes(mine, always=True)]}

    def build_info(self):
        """Whether this process's library was compiled in this checkout state or reused as shipped."""
        from kzg_snark_amd import build as B
        st = B._read_stamp()
        return {"library": os.path.relpath(self.native.LIB_PATH, ROOT), "source_hash": st.get("source_hash", "")[:16],
                "matches_sources": st.get("source_hash") == B.source_hash(), "hipcc": st.get("hipcc")}

    def affine(self, pt):
        q = self.kzg._g1.normalize(pt)
        return (int(q[0]), int(q[1]))

    def g1_times(self, s):
        return self.affine(self.kzg.multiply(self.kzg.G1, s % self.r))

    def point(self, xy, inf=0):
        import numpy as np
        if inf:
            return self.kzg.Z1
        v = self.native.limbs_to_ints(np.asarray(xy).reshape(2, self.ctx.fp_limbs))
        return (v[0], v[1], 1)

    def uniform(self, shape_prefix, seed):
        """uniform-looking canonical Fr elements (< 2^251 < r) generated on the device."""
        torch = self.torch
        g = torch.Generator(device=self.dev).manual_seed(seed)
        t = torch.randint(0, 1 << 62, (*shape_prefix, 4), generator=g, dtype=torch.int64, device=self.dev)
        t[..., 3] >>= 3
        return t


def section_open(env, srs, n):
    import numpy as np
    """KZG.open (kzg.py:122-159) of k polynomials of n coefficients: combine with xi^(i+1), evaluate at z,
    divide by (X - z) -- csrc/poly.hip -- then one MSM."""
    args, ctx, nat = env.args, env.ctx, env.native
    k = args.open_k
    polys = env.uniform((k, n), 0x6f70656e + env.rank)
    lens = [n - i for i in range(k)]                     # ragged like the provers' n+2..n+6 (shorter ones read as zero-padded)
    zw, xw = nat.int_to_words(Z_PT % env.r), nat.int_to_words(XI_PT % env.r)
    env.torch.cuda.synchronize(env.dev)
    for _ in range(2):
        ctx.open(srs, polys.data_ptr(), lens, n, zw, xw, device=True)
    ctx.prof_enable(True)
    ctx.prof_reset()
    iters = max(5, min(args.steps, 20))
    env.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        xy, inf, ev = ctx.open(srs, polys.data_ptr(), lens, n, zw, xw, device=True)
    env.barrier()
    elapsed = env.max_over_ranks(time.perf_counter() - t0)
    poly_ms, poly_cnt = ctx.prof_read("open_poly")
    ctx.prof_enable(False)
    # the same openings through the pipelined entry point (kzg_open_device_async + kzg_commit_flush): the witness
    # MSMs share the commit pipeline, several openings in flight
    L = ctx.fp_limbs
    # as many openings as the headline loop commits polynomials (steps x batch): a short loop is mostly pipeline fill
    # and drain (the first polynomial stage and the last reduce stage, ~1 ms, over `n_async` x 2.2 ms)
    n_async = max(iters, args.steps * max(1, args.batch))
    outs = [(np.zeros(2 * L, dtype=np.uint64), np.zeros(1, dtype=np.uint8), np.zeros(4, dtype=np.uint64))
            for _ in range(n_async)]
    for o in outs[:2]:
        ctx.open_device_async(srs, polys.data_ptr(), lens, n, zw, xw, *o)
    ctx.commit_flush()
    env.barrier()
    t0 = time.perf_counter()
    for o in outs:
        ctx.open_device_async(srs, polys.data_ptr(), lens, n, zw, xw, *o)
    ctx.commit_flush()
    env.barrier()
    elapsed_async = env.max_over_ranks(time.perf_counter() - t0)
    ok_async = all(np.array_equal(o[0], xy) and o[1][0] == inf[0] and np.array_equal(o[2], ev) for o in outs)
    # trapdoor: proof == ((P(tau) - P(z)) / (tau - z)) G1 with P = sum xi^(i+1) p_i
    r, tau, z, xi = env.r, TAU % env.r, Z_PT % env.r, XI_PT % env.r
    Pt = sum(pow(xi, i + 1, r) * ctx.poly_eval(lens[i], polys[i].data_ptr(), tau) for i in range(k)) % r
    Pz = int.from_bytes(ev.tobytes(), "little")
    ok = (not inf[0]) and env.affine(env.point(xy)) == env.g1_times((Pt - Pz) * pow((tau - z) % r, -1, r))
    ok_ev = Pz == sum(pow(xi, i + 1, r) * ctx.poly_eval(lens[i], polys[i].data_ptr(), z) for i in range(k)) % r
    alg_bytes = (k + 1) * n * 32                            # read k polynomials, write the quotient
    avg_s = poly_ms / max(poly_cnt, 1) * 1e-3
    # HBM bytes per opening from the PMC passes of tools/profile_open.sh (profiles/counters.json "open_poly"); only
    # quoted for the configuration they were collected on
    traffic = None
    cfile = os.path.join(ROOT, "profiles", "counters.json")
    if os.path.exists(cfile):
        ent = json.load(open(cfile)).get("open_poly") or {}
        if ent.get("k") == k and ent.get("log_n") == n.bit_length() - 1:
            traffic = ent.get("hbm_bytes")
    return {
        "value": env.world * iters / elapsed, "unit": "opens/s", "k": k, "log_n": n.bit_length() - 1,
        "ms_per_open": elapsed / iters * 1e3,
        "poly_stage_ms": avg_s * 1e3,
        "pipelined": {"value": env.world * n_async / elapsed_async, "unit": "opens/s",
                      "ms_per_open": elapsed_async / n_async * 1e3, "openings": n_async,
                      "entry": "kzg_open_device_async + kzg_commit_flush"},
        "verified": {"proof_trapdoor": bool(ok), "combined_eval": bool(ok_ev), "pipelined_equals_synchronous": bool(ok_async)},
        "poly_stage_spans": int(poly_cnt),                 # == iters: ONE open_poly span per opening
        "roofline": {"kernel": "open_poly: tile_combine_kernel + tile_fill_kernel (csrc/poly.hip), one span per opening",
                     "bound": "hbm",
                     "achieved": alg_bytes / avg_s / 1e9 if avg_s > 0 else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": alg_bytes / avg_s / 1e9 / HBM_PEAK_GBPS if avg_s > 0 else None, "traffic": traffic,
                     "algorithmic_bytes_per_open": alg_bytes},
    }, bool(ok and ok_ev and ok_async)


def section_range(env, log_n=None, steps=None):
    """BASELINE config 4 (log_n = None: --range-log-n, 24), and the same step at the metric's own degree (log_n = 20:
    strong scaling of ONE commit + open over the ranks): k polynomials of 2^log_n coefficients and the commitment key split by contiguous
    coefficient range over the ranks.  Step = commit of polynomial 0 + open of all k:
    every rank runs a whole local MSM per operation; what crosses ranks is one field element (open) and one
    point per rank per operation, all-gathered as fixed-size records and added on the host by every rank."""
    from kzg_snark_amd.sharding import DistributedCommitter, all_gather_bytes, range_of
    args, ctx, nat, kzg, r = env.args, env.ctx, env.native, env.kzg, env.r
    world, rank = env.world, env.rank
    log_n, k = (args.range_log_n if log_n is None else log_n), args.open_k
    n_steps = args.range_steps if steps is None else steps
    n = 1 << log_n
    lo, hi = range_of(rank, world, n)
    m = hi - lo
    tau, z, xi = TAU % r, Z_PT % r, XI_PT % r
    tw, zw, xw = nat.int_to_words(tau), nat.int_to_words(z), nat.int_to_words(xi)
    t0 = time.perf_counter()
    cshard = ctx.srs_generate(tw, m, start=lo)                      # tau^lo .. tau^(hi-1)
    start = 0 if rank == 0 else lo - 1
    oshard = ctx.srs_generate(tw, hi - 1 - start, start=start)      # key slice of the quotient's coefficients
    ctx.synchronize()
    t_srs = time.perf_counter() - t0
    sl = env.uniform((k, m), 0x72616e67 + rank)
    lens = [m] * k
    env.torch.cuda.synchronize(env.dev)

    def commit_fn(_polys):
        xy, inf = ctx.commit_device(cshard, sl.data_ptr(), [m], m)
        return [env.point(xy[0], inf[0])]

    def begin_fn():
        return nat.limbs_to_ints(ctx.open_shard_begin(sl.data_ptr(), lens, m, zw, xw).reshape(1, 4))[0]

    def finish_fn(carry, first):
        xy, inf, ev = ctx.open_shard_finish(oshard, zw, nat.int_to_words(carry), first)
        return env.point(xy, inf[0]), (nat.limbs_to_ints(ev.reshape(1, 4))[0] if first else None)

    dc = DistributedCommitter(commit_fn, kzg.add, kzg.Z1, sum_fn=lambda pts: nat.g1_sum(args.curve, pts))
    import numpy as np
    c_xy, c_inf = np.zeros((1, 2 * ctx.fp_limbs), dtype=np.uint64), np.zeros(1, dtype=np.uint8)

    def start_commit():                                    # the partial commitment rides the pipeline ...
        ctx.commit_device_async(cshard, sl.data_ptr(), [m], m, c_xy, c_inf)

    def collect_commit():                                  # ... and is complete once the opening's MSM has drained it
        ctx.commit_flush()
        return env.point(c_xy[0], c_inf[0])

    def step():
        return dc.commit_and_open_range(start_commit, collect_commit, begin_fn, finish_fn, z, r, n)

    for _ in range(2):
        step()
    env.barrier()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        commitment, (proof, ev) = step()
    env.barrier()
    elapsed = env.max_over_ranks(time.perf_counter() - t0)
    # trapdoor identities, evaluated shard-wise on the devices: p_i(tau) = sum_g tau^lo_g * slice_(g,i)(tau)
    mine = b"".join((ctx.poly_eval(m, sl[i].data_ptr(), tau) * pow(tau, lo, r) % r).to_bytes(32, "little")
                    for i in range(k))
    p_tau = [0] * k
    for blob in all_gather_bytes(mine):
        for i in range(k):
            p_tau[i] = (p_tau[i] + int.from_bytes(blob[32 * i:32 * i + 32], "little")) % r
    ok_commit = env.affine(commitment) == env.g1_times(p_tau[0])
    P_tau = sum(pow(xi, i + 1, r) * p_tau[i] for i in range(k)) % r
    ok_open = env.affine(proof) == env.g1_times((P_tau - ev) * pow((tau - z) % r, -1, r))
    del cshard, oshard, sl
    out = {
        "metric": f"KZG commit + open of k = {k} per second, degree-2^{log_n} polynomials sharded by coefficient range",
        "value": n_steps / elapsed, "unit": "commit+open/s", "n_gpus": world, "steps": n_steps,
        "ms_per_step": elapsed / n_steps * 1e3, "scaling": "strong",
        "log_n": log_n, "k": k, "coefficients_per_rank": m,
        "exchange": "per step: one 32-byte field element per rank, then one record of two 97-byte G1 points "
                    "(+ P(z)) per rank, all-gathered as uint8 tensors and added on the host; the two local MSMs "
                    "share the commit pipeline",
        "verified": {"commit_trapdoor": bool(ok_commit), "open_trapdoor": bool(ok_open)},
        "srs_setup_s": t_srs,
    }
    return out, bool(ok_commit and ok_open)


def section_plonk(env):
    """BASELINE config 5: index -> prove -> verify of a synthetic mul/add-chain circuit of 2^20 gates with the
    prover's polynomials resident on the GPU (kzg_snark_amd/plonk_device.py; plonk/prover.py:24-212).  The value
    is the wall time of a prover round with the witness handed over as limbs (what a native witness generator
    emits), median of three after the first; the verifier must accept the last proof."""
    torch = env.torch
    from kzg_snark_amd import plonk, plonk_device
    from kzg_snark_amd.sharding import ProofSharding
    n = 1 << env.args.plonk_log_n
    # N > 1: ONE proof over the ranks -- the commitments of a round and the two openings dealt round-robin against
    # the replicated key, vector work replicated (sharding.ProofSharding); every rank ends with the same proof
    pow2 = env.world & (env.world - 1) == 0
    sharded = env.collective and (env.args.plonk_mode == "sharded" or (env.args.plonk_mode == "auto" and pow2 and env.world > 1))
    sh = ProofSharding(shard_vectors=sharded) if env.collective else None
    prev = torch.cuda.current_stream(env.dev)
    t = {}
    t0 = time.perf_counter()
    qM, qL, qR, qO, qC, perm, x, w = plonk.synthetic_circuit(n, env.kzg.Fq, seed=env.args.plonk_log_n)
    t["circuit_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx = plonk_device.DeviceIndexer(env.args.curve)
    ipk, ivk = idx.preprocess(qM, qL, qR, qO, qC, perm, tau=0x706c6f6e6b if sh else None)   # one key on every rank
    t["index_s"] = time.perf_counter() - t0
    from kzg_snark_amd import plonk_sharded
    prv = plonk_sharded.make_prover(env.args.curve, idx.alg, sh)
    sharded = type(prv).__name__ == "ShardedProver"
    env.barrier()
    t0 = time.perf_counter()
    proof = prv.prove(ipk, x, w)
    t["prove_first_s"] = time.perf_counter() - t0
    w_limbs = env.native.ints_to_limbs([int(v) for v in w])
    rounds = []
    for _ in range(3):
        env.barrier()
        t0 = time.perf_counter()
        proof = prv.prove(ipk, x, w_limbs)
        rounds.append(env.max_over_ranks(time.perf_counter() - t0))
    t0 = time.perf_counter()
    ok = bool(plonk.Verifier(env.args.curve).verify(ivk, x, proof))
    t["verify_s"] = time.perf_counter() - t0
    same = True
    if sh is not None:      # every rank must hold the same proof
        import hashlib
        from kzg_snark_amd.sharding import all_gather_bytes
        digest = hashlib.sha256(repr(sorted((k, tuple(int(c) for c in v)) for k, v in
                                            list(proof["commitments"].items()) + list(proof["kzg_proofs"].items()))
                                     ).encode()).digest()
        same = len(set(all_gather_bytes(digest))) == 1
    exch = (prv.exchanges // 4, prv.tf.exchanges // 4) if sharded else None       # four proofs were made
    del ipk, prv, idx
    torch.cuda.set_stream(prev)
    torch.cuda.empty_cache()
    out = {"metric": "PLONK prover round, 2^%d gates, %s, device-resident" % (env.args.plonk_log_n, env.args.curve),
           "value": sorted(rounds)[1] * 1e3, "unit": "ms", "higher_is_better": False, "gates": n, "n_gpus": env.world,
           "rounds_ms": [v * 1e3 for v in rounds], **t,
           "verified": {"verifier_accepts": ok, **({"same_proof_on_every_rank": same} if sh else {})}}
    if sharded:
        out["sharding"] = ("one proof over the ranks, vectors split (plonk_sharded.ShardedProver): range / transposed "
                           "shards, distributed transforms, partial commitments and openings against key shards")
        out["exchanges_per_proof"] = {"record_gathers": exch[0], "tensor_collectives": exch[1]}
    elif sh is not None:
        out["sharding"] = ("one proof over the ranks: 3 + 1 + 3 commitments and the 2 openings dealt round-robin, "
                           "replicated key and vector work; %d exchanges per proof (one all-gather of 97-byte records "
                           "per round + the blinders)" % (sh.exchanges // 4))
    return out, ok and same


def section_dist_ntt(env, log_n=None):
    """The four-step transform of 2^log_n elements (None: --range-log-n, 24; 20: the metric's own degree) sharded over the ranks (sharding.DistributedNTT): forward
    in natural order (three all-to-alls), inverse into the transposed layout (two).  Every rank also runs
    the whole transform alone and compares its shard."""
    from kzg_snark_amd.sharding import DistributedNTT, GpuNttOps, transposed_index
    args, ctx, nat, torch = env.args, env.ctx, env.native, env.torch
    world, rank = env.world, env.rank
    log_n = args.range_log_n if log_n is None else log_n
    k1 = (log_n + 1) // 2
    N1, N2 = 1 << k1, 1 << (log_n - k1)
    if world & (world - 1) or N1 % world or N2 % world or log_n <= 12:
        return {"skipped": f"world = {world} must be a power of two dividing 2^{log_n - k1}"}, True
    n = 1 << log_n
    r, gen = (R_BLS, 7) if args.curve == "bls12_381" else (R_BN, 5)
    ww = nat.int_to_words(root_of_unity(r, gen, n))
    full = env.uniform((n,), 0x646e7474)                 # the same on every rank
    lo, hi = rank * n // world, (rank + 1) * n // world
    out = {"log_n": log_n, "n_gpus": world, "backend": args.backend}
    ok_all = True
    for name, inverse, layout in (("forward_natural", False, "natural"), ("inverse_transposed", True, "transposed")):
        d = DistributedNTT(GpuNttOps(ctx, log_n, ww, inverse))
        for _ in range(2):
            res = d.transform(full[lo:hi].clone(), log_n, layout=layout)
        iters = 5
        xs = [full[lo:hi].clone() for _ in range(iters)]
        env.barrier()
        t0 = time.perf_counter()
        for i in range(iters):
            res = d.transform(xs[i], log_n, layout=layout)
        env.barrier()
        elapsed = env.max_over_ranks(time.perf_counter() - t0)
        ref = full.clone()
        ctx.ntt_device(ref.data_ptr(), log_n, ww, inverse, 1)
        torch.cuda.synchronize(env.dev)
        if layout == "natural":
            ok = bool(torch.equal(res, ref[lo:hi]))
        else:
            idx = torch.arange(hi - lo, device=env.dev)
            gidx = (idx % N2) * N1 + rank * (N1 // world) + idx // N2
            assert int(gidx[1]) == transposed_index(log_n, world, rank, 1)
            ok = bool(torch.equal(res, ref[gidx]))
        flag = torch.tensor([1 if ok else 0], device=env.dev if args.backend == "nccl" else "cpu")
        if env.collective:
            env.dist.all_reduce(flag, op=env.dist.ReduceOp.MIN)
        ok = bool(int(flag.item()))
        ok_all &= ok
        out[name] = {"ms": elapsed / iters * 1e3, "elements_per_s": n * iters / elapsed,
                     "all_to_alls": 3 if layout == "natural" else 2, "verified": ok}
        del xs, ref
    del full
    return out, ok_all


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        return launch(args, argv)
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: WORLD_SIZE = {world} but --gpus {args.gpus}: the launcher and the label disagree",
                  file=sys.stderr)
        return 2
    if args.rehearse_launch:
        return rehearse_launch(args, world, rank)

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    if args.one_device:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no GPU (local rank {local_rank}, "
                         f"{torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)              # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local_rank)
    rehearse = args.rehearse_collectives and world == 1
    if world > 1 or rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            from kzg_snark_amd import sharding
            sharding.FORCE_COLLECTIVES = True
        # RCCL: name the device of this rank (set above) instead of letting the backend guess it from the global rank
        dist.init_process_group(args.backend, rank=rank, world_size=world,
                                **({"device_id": dev} if args.backend == "nccl" else {}))

    from kzg_snark_amd import _native
    ctx = _native.Context(args.curve, device=local_rank)
    # a dedicated (non-null) torch stream shared with the library: torch copies and the
    # library's kernels are then ordered on one queue
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.bind_torch_stream(stream)
    env = Env(args, torch, dist, ctx, dev, world, rank, _native)

    def finish(code):
        if env.collective:
            dist.barrier()
            dist.destroy_process_group()
        return code

    if args.mode == "range":
        out, ok = section_range(env)
        if rank == 0:
            out.update({"warmup": 2, "higher_is_better": True, "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
                        "config": {"workload": f"degree-2^{args.range_log_n} commit + open of {args.open_k}, {args.curve}, "
                                               f"key and polynomials split by coefficient range over {world} rank(s)"}})
            print(json.dumps(out), flush=True)
        return finish(0 if ok else 1)

    r, gen = (R_BLS, 7) if args.curve == "bls12_381" else (R_BN, 5)
    fp_bytes = 48 if args.curve == "bls12_381" else 32
    log_n = args.log_n
    n = 1 << log_n
    omega = root_of_unity(r, gen, n)
    w_words = _native.int_to_words(omega)

    # synthetic workload: SRS [tau^i G1] generated on the device; uniform Fr evaluations (seeded per rank)
    tau = TAU % r
    t0 = time.perf_counter()
    srs = ctx.srs_generate(_native.int_to_words(tau), n)
    ctx.synchronize()
    t_srs = time.perf_counter() - t0
    B = args.batch
    evals = env.uniform((B, n), 0x6b7a + rank)
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

    for i in range(args.warmup):
        step(i)
    ctx.commit_flush()
    ctx.prof_enable(True)
    ctx.prof_reset()
    env.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    ctx.commit_flush()                                      # every result is on the host before the clock stops
    env.barrier()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    elapsed = env.max_over_ranks(elapsed)
    spans_main = {name: ctx.prof_read(name) for name in
                  ("msm_partition1", "msm_partition2", "msm_order", "msm_accumulate", "msm_finalize", "msm_reduce")}
    mhz_pipelined = ctx.prof_read("msm_accumulate_shader_mhz")[0]

    # ---- the timed loop's own results, checked (outside the timed region): the coefficients of the last
    # step are still in its work buffer; commit(ck, p) must be p(tau) G1 for each of its B polynomials,
    # and no earlier step may have produced infinity or the zero record
    last = args.steps - 1
    ok_headline = all(int(inf.sum()) == 0 and xy.any(axis=1).all() for xy, inf in results.values())
    if args.steps > 0:
        xy_last, inf_last = results[last]
        for p in range(B):
            p_tau = ctx.poly_eval(n, works[last & 1][p].data_ptr(), tau)
            ok_headline &= (not inf_last[p]) and env.affine(env.point(xy_last[p])) == env.g1_times(p_tau)
    flag = torch.tensor([1 if ok_headline else 0], device=dev if args.backend == "nccl" else "cpu")
    if env.collective:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ok_headline = bool(int(flag.item()))

    # The NTT half of the metric, timed alone (inside the pipelined loop above its kernels share the
    # GPU with the previous polynomial's bucket reduction, which inflates their event times).
    ctx.prof_enable(True)
    ctx.prof_reset()
    ntt_iters = max(args.steps, 10)
    for i in range(ntt_iters):
        ctx.ntt_device(works[0].data_ptr(), log_n, w_words, bool(i & 1) ^ True, B)
    env.barrier()
    ntt_alone = ctx.prof_read("ntt_pass")
    mhz_ntt = ctx.prof_read("ntt_pass_shader_mhz")[0]
    # The accumulate kernel alone: in the pipelined loop it deliberately shares every SIMD with the
    # next polynomial's prep and the previous one's reduce stage, so its span there is the pipeline
    # period.  One commit at a time (flush after each) gives the kernel's own duration.
    ctx.prof_reset()
    for p in range(0 if args.no_isolated else B):
        xy1 = np.zeros((1, 2 * fp_limbs), dtype=np.uint64)
        inf1 = np.zeros(1, dtype=np.uint8)
        ctx.commit_device_async(srs, works[0].data_ptr() + p * n * 32, [n], n, xy1, inf1)
        ctx.commit_flush()
    env.barrier()
    acc_alone = ctx.prof_read("msm_accumulate")
    mhz_alone = ctx.prof_read("msm_accumulate_shader_mhz")[0]
    spans_alone = {name: ctx.prof_read(name) for name in
                   ("msm_partition1", "msm_partition2", "msm_order", "msm_accumulate", "msm_finalize", "msm_reduce")}
    ctx.prof_enable(False)

    sections, ok_sections = {}, True

    def run_section(name, fn, *fargs):
        """The sections ride on the headline line; one that raises is reported as such in its place (and on
        stderr, and in the exit code) instead of taking the headline measurement down with it."""
        nonlocal ok_sections
        try:
            sections[name], ok = fn(*fargs)
        except Exception as e:   # noqa: BLE001 -- whatever it is, it is reported, not swallowed
            import traceback
            traceback.print_exc()
            sections[name], ok = {"error": f"{type(e).__name__}: {e}", "verified": False}, False
        ok_sections &= ok

    if args.mode == "all" and not args.no_open:
        run_section("open", section_open, env, srs, n)
    del srs, works, evals
    torch.cuda.empty_cache()
    if args.mode == "all" and not args.no_range:
        run_section("range_mode", section_range, env)
    if args.mode == "all" and env.collective and not args.no_dist_ntt:
        run_section("distributed_ntt", section_dist_ntt, env)
    # strong scaling at the metric's own degree (BASELINE.json: "commits/sec + NTT elements/sec at degree 2^20, 1/2/4/8
    # MI355X"): ONE degree-2^log_n commit + open split by coefficient range, and ONE 2^log_n transform over the
    # all-to-alls -- the headline `value` above is batch mode (independent polynomials per rank, weak scaling)
    if args.mode == "all" and env.collective and args.range_log_n != log_n:
        tag = "2p%d" % log_n
        if not args.no_range:
            run_section("range_mode_" + tag, section_range, env, log_n, max(args.range_steps, 20))
        if not args.no_dist_ntt:
            run_section("distributed_ntt_" + tag, section_dist_ntt, env, log_n)
    if args.mode == "all" and not args.no_plonk:
        run_section("plonk_round", section_plonk, env)

    coll_info = env.collectives_info()      # a collective: every rank takes part
    # the NTT half of the metric over all ranks: every rank ran the same loop on its own GPU; the slowest rank's time
    ntt_pass_s_all = env.max_over_ranks(ntt_alone[0] / max(ntt_alone[1], 1) * 1e-3)
    spans = dict(spans_main)
    spans["ntt_pass"] = (ntt_alone[0] * args.steps / ntt_iters, ntt_alone[1] * args.steps // ntt_iters)
    if rank == 0:
        acc_ms, acc_cnt = spans["msm_accumulate"]
        acc_avg_s = (acc_ms / max(acc_cnt, 1)) * 1e-3
        msm_bytes = n * (32 + 2 * fp_bytes)                 # scalars + affine points, per commit
        ntt_ms, ntt_cnt = ntt_alone
        # one launch covers the whole batch; two launches (passes) per transform above 2^12
        ntt_per_transform_s = ntt_pass_s_all * (2 if log_n > 12 else 1) / B
        ntt_bytes = 2 * n * 32                              # read + write every element once
        acc_alone_s = acc_alone[0] / max(acc_alone[1], 1) * 1e-3
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
            "dtype": DTYPE,
            "data": "synthetic",
            "config": {"workload": f"degree-2^{log_n} INTT + KZG commit, {args.curve}, 2^{log_n}-point SRS, "
                                   f"uniform Fr scalars, batch of {B} polynomials per GPU per step",
                       "log_n": log_n, "curve": args.curve, "batch": B, "window_bits": 20 if n >= (1 << 18) else 16,
                       "sharding": "independent polynomials per rank, replicated SRS"},
            "verified": {"last_step_commit_trapdoor": ok_headline},
            # what the process group really was (a SCALE record then shows that RCCL saw N ranks, one GPU each)
            "collectives": coll_info,
            "build": env.build_info(),
            # whole job: every rank transforms its own polynomials (slowest rank's time per transform)
            "ntt_elements_per_s": world * n / ntt_per_transform_s if ntt_per_transform_s > 0 else None,
            "ntt_ms": ntt_per_transform_s * 1e3,
            "kernel_ms_per_commit": {k: (v[0] / (args.steps * B)) for k, v in spans.items()},
            # the same spans with ONE commit in flight (nothing else on the GPU): each stage's own duration
            "kernel_ms_per_commit_isolated": {k: (v[0] / max(v[1], 1)) for k, v in spans_alone.items()},
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
                    "avg_launch_ms": acc_alone_s * 1e3,
                    "achieved": msm_bytes / acc_alone_s / 1e9 if acc_alone_s > 0 else None,
                    "frac": msm_bytes / acc_alone_s / 1e9 / HBM_PEAK_GBPS if acc_alone_s > 0 else None,
                },
                "note": "in the timed loop the persistent accumulate kernel runs beside prep(p+1) and reduce(p-1); "
                        "its span there is the pipeline period.  The kernel is VALU-issue bound, not HBM bound: see "
                        "`limiter`",
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
        counters_file = os.path.join(ROOT, "profiles", "counters.json")
        if os.path.exists(counters_file):
            # rocprofv3 --pmc passes of this build (tools/profile_bench.sh -> tools/summarize_profile.py):
            # HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, gfx950 corrections applied) and VALU
            # wave-instructions per launch.  The second object names the limiter the counters show.
            cj = json.load(open(counters_file))
            for key, kern, dur_s in (("roofline", "msm_accumulate_kernel", acc_alone_s),
                                     ("roofline_ntt", "ntt_pass_kernel_per_transform", ntt_per_transform_s)):
                ent = cj.get(kern) or {}
                out[key]["traffic"] = ent.get("hbm_bytes")
                if ent.get("valu_wave_instructions") and dur_s > 0:
                    ach = ent["valu_wave_instructions"] / dur_s / 1e9
                    out[key]["limiter"] = {"bound": "valu", "achieved": ach, "peak": VALU_PEAK_GINSTR,
                                           "unit": "G wave-instr/s", "frac": ach / VALU_PEAK_GINSTR,
                                           "valu_wave_instructions_per_launch": ent["valu_wave_instructions"],
                                           "hbm_achieved_GBps_from_traffic": (ent["hbm_bytes"] / dur_s / 1e9)
                                           if ent.get("hbm_bytes") else None,
                                           "source": cj.get("source")}
                    if key == "roofline_ntt" and mhz_ntt > 0:
                        peak_at_clock = VALU_PEAK_GINSTR * mhz_ntt / SHADER_PEAK_MHZ
                        out[key]["limiter"].update({"shader_mhz": mhz_ntt, "peak_at_measured_clock": peak_at_clock,
                                                    "frac_at_measured_clock": ach / peak_at_clock})
                    if key == "roofline" and mhz_alone > 0:
                        # the shader clock is power-managed: the kernel reports the one it actually ran at
                        # (s_memtime / s_memrealtime, measured live), and the issue peak at THAT clock
                        peak_at_clock = VALU_PEAK_GINSTR * mhz_alone / SHADER_PEAK_MHZ
                        out[key]["limiter"].update({"shader_mhz_isolated": mhz_alone,
                                                    "shader_mhz_pipelined": mhz_pipelined or None,
                                                    "peak_at_measured_clock": peak_at_clock,
                                                    "frac_at_measured_clock": ach / peak_at_clock})
        out.update(sections)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.curve, log_n, omega)
            out["cpu_baseline_optimised"] = cpu_baseline_optimised(args.curve, log_n, omega)
        print(json.dumps(out), flush=True)
    ok_all = ok_headline and ok_sections
    if not ok_all and rank == 0:
        print("bench.py: a result failed verification (see the `verified` flags)", file=sys.stderr)
    return finish(0 if ok_all else 1)


if __name__ == "__main__":
    sys.exit(main() or 0)
