"""World-size-2 gloo tests (CPU) of the multi-GPU exchange logic (kzg_snark_amd/sharding.py).
The per-rank commit is the oracle's CPU commit standing in for the GPU MSM -- the exchange
(round-robin ownership, all-gather, host-side point addition) is what is under test."""
import os
import random
import socket

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import py_oracle as O


import datetime

_PG_TIMEOUT = datetime.timedelta(seconds=240)        # a rank whose peer died gives up instead of waiting half an hour


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(target, world, timeout, extra=()):
    """Spawn `world` ranks of `target(rank, world, port, queue, *extra)` and collect one result each.  Whatever goes
    wrong -- a rank that raised leaves its peers blocked in a gloo collective, a queue that stays empty -- every
    child is terminated (then killed) before the error travels on: no process is left holding the GPU or the port,
    and pytest does not hang joining non-daemon children."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q, *extra)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = []
        for _ in range(world):
            res.append(q.get(timeout=timeout))
            if res[-1][1] != "ok":                      # its peers may never return from their collective
                break
        for p in procs:
            p.join(timeout=60 if len(res) == world and all(v == "ok" for _, v in res) else 1)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    return res


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=_PG_TIMEOUT)
    try:
        from kzg_snark_amd import curve as C
        from kzg_snark_amd.sharding import DistributedCommitter, range_of
        cv = O.BN254
        G = C.g1_group(C.CURVES["bn254"])
        tau = 0xabcdef
        n = 13
        full_ck = O.setup(n - 1, tau, cv)
        rng = random.Random(5)
        polys = [[rng.randrange(cv.r) for _ in range(rng.randrange(1, n + 1))] for _ in range(4)] + [[0, 0]]

        def to_pt(p):
            a = O.normalize(p, cv)
            return (1, 1, 0) if a is None else (a[0], a[1], 1)

        # batch mode: replicated key, polynomials dealt round-robin
        dc = DistributedCommitter(lambda ps: [to_pt(c) for c in O.commit(full_ck, ps, cv)], G.add, G.Z)
        got = dc.commit_batch(polys)
        want = [to_pt(c) for c in O.commit(full_ck, polys, cv)]
        assert got == want, "batch mode"

        # range mode: key and coefficients sharded by contiguous range
        lo, hi = range_of(rank, world, n)
        shard_ck = full_ck[lo:hi]
        p = [rng.randrange(cv.r) for _ in range(n)]
        dc2 = DistributedCommitter(lambda ps: [to_pt(c) for c in O.commit(shard_ck, ps, cv)], G.add, G.Z)
        got = dc2.commit_range(p[lo:hi])
        assert got == to_pt(O.commit(full_ck, [p], cv)[0]), "range mode"
        # the same with the partial points added by the library's host group law (kzg_g1_sum) instead of one by one
        from kzg_snark_amd import _native
        dc3 = DistributedCommitter(lambda ps: [to_pt(c) for c in O.commit(shard_ck, ps, cv)], G.add, G.Z,
                                   sum_fn=lambda pts: _native.g1_sum("bn254", pts))
        assert dc3.commit_range(p[lo:hi]) == got, "range mode, sum_fn"
        assert got == to_pt(O.commit_trapdoor(p, tau, cv))

        # range-mode open: slice evaluation, carry exchange, quotient slice against a key shard that
        # starts at lo - 1 (the oracle stands in for kzg_open_shard_begin / _finish)
        r = cv.r
        polys3 = [[rng.randrange(r) for _ in range(m)] for m in (n, 9, 4)]
        z, xi = rng.randrange(r), rng.randrange(r)
        comb = O.combine(polys3, xi, r) + [0] * n
        comb = comb[:n]
        local = comb[lo:hi]

        def begin():
            return O.poly_eval(local, z, r)

        def finish(carry, first):
            ext = local + [carry]
            S = [0] * (len(ext) + 1)
            for j in range(len(ext) - 1, -1, -1):
                S[j] = (ext[j] + z * S[j + 1]) % r
            if first:
                vec, start, ev = S[1:len(local)], 0, S[0]
            else:
                vec, start, ev = S[0:len(local)], lo - 1, None
            key = full_ck[start:start + len(vec)]
            return to_pt(O.commit(key, [vec], cv)[0]) if vec else (1, 1, 0), ev

        proof, ev = dc2.open_range(begin, finish, z, r, n)
        want, pz = O.open_(full_ck, polys3, z, xi, cv)
        assert proof == to_pt(want) and ev == pz, "range-mode open"

        # the fused step (one polynomial committed and k opened, two exchanges): same results
        box = {}
        com, (proof2, ev2) = dc2.commit_and_open_range(
            lambda: box.setdefault("c", to_pt(O.commit(shard_ck, [p[lo:hi]], cv)[0])), lambda: box["c"],
            begin, finish, z, r, n)
        assert com == to_pt(O.commit(full_ck, [p], cv)[0]) and proof2 == proof and ev2 == ev, "fused range step"
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _oracle_backends():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import oracle_backends
    return oracle_backends


def _ntt_worker(rank, world, port, q):
    import numpy as np
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=_PG_TIMEOUT)
    try:
        from kzg_snark_amd.sharding import DistributedNTT
        cv = O.BLS12_381
        log_n = 7
        n = 1 << log_n
        rng = random.Random(99)
        x = [rng.randrange(cv.r) for _ in range(n)]
        for w, inverse in ((cv.root_of_unity(n), False), (cv.root_of_unity(n), True), (rng.randrange(2, cv.r), False)):
            want = O.ifft_ff(x, w, cv.r) if inverse else O.fft_ff(x, w, cv.r)
            lo, hi = rank * n // world, (rank + 1) * n // world
            buf = b"".join(v.to_bytes(32, "little") for v in x[lo:hi])
            xl = torch.from_numpy(np.frombuffer(buf, dtype="<i8").reshape(hi - lo, 4).copy())
            OB = _oracle_backends()
            d = DistributedNTT(OB.OracleNttOps(log_n, w, cv.r, inverse))
            got = OB.ints_of(d.transform(xl.clone(), log_n))
            assert got == want[lo:hi], (inverse, "distributed NTT shard mismatch")
            # two all-to-alls: the result stays in the row pass's order
            from kzg_snark_amd.sharding import transposed_index
            got_t = OB.ints_of(d.transform(xl.clone(), log_n, layout="transposed"))
            assert got_t == [want[transposed_index(log_n, world, rank, i)] for i in range(hi - lo)], "transposed layout"
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_distributed_ntt():
    """The all-to-all choreography of the multi-GPU four-step NTT (natural order in; natural order
    out with three exchanges or the transposed layout with two; arbitrary w, inverse) with the
    oracle as the local transform."""
    res = _run_ranks(_ntt_worker, 2, 180)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_two_rank_exchange():
    res = _run_ranks(_worker, 2, 120)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_range_partition():
    from kzg_snark_amd.sharding import range_of, round_robin
    for n in (1, 7, 8, 1 << 20, (1 << 20) + 6):
        for world in (1, 2, 3, 8):
            parts = [range_of(r, world, n) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in parts) - min(h - l for l, h in parts) <= 1
    assert round_robin(5, 2) == [0, 1, 0, 1, 0]


def _golden(name):
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)))


def _check_frozen_proof(proof, gp):
    """proof dict == tests/golden/plonk_proof_n16.json, bit for bit"""
    want = gp["proof"]
    for k, v in want["commitments"].items():
        assert tuple(int(c) for c in proof["commitments"][k]) == (int(v[0], 16), int(v[1], 16), v[2]), k
    for k, v in want["evaluations"].items():
        assert int(proof["evaluations"][k]) == int(v, 16), k
    for k, v in want["kzg_proofs"].items():
        assert tuple(int(c) for c in proof["kzg_proofs"][k]) == (int(v[0], 16), int(v[1], 16), v[2]), k


def _plonk_worker(rank, world, port, q, device):
    """One rank of a PLONK proof dealt over `world` ranks (sharding.ProofSharding; BASELINE config 5 on several
    GPUs).  device=False: host prover, the oracle standing in for the engine's commit / open / INTT (CPU);
    device=True: the device prover on cuda:0 (every rank on the test box's one GPU)."""
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=_PG_TIMEOUT)
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        if here not in sys.path:
            sys.path.insert(0, here)
        import test_plonk as TP
        from kzg_snark_amd import plonk
        from kzg_snark_amd.sharding import ProofSharding
        gp = _golden("plonk_proof_n16.json")
        circuit = TP.fixture_instance()
        curve, tau = gp["curve"], int(gp["tau"], 16)
        blinders = [int(v, 16) for v in gp["blinders"]]
        if not device:
            from kzg_snark_amd.field import PolynomialRing
            plonk.fft_ff_interpolation = lambda values, g, F: PolynomialRing(F, "X")(
                O.fft_ff_interpolation([int(v) for v in values], int(g), F.p))
            sh = ProofSharding()
            idx, prv = plonk.Indexer(curve), plonk.Prover(curve, sharding=sh)
            idx.kzg = prv.kzg = TP.oracle_backed(curve)
            ipk, ivk = idx.preprocess(*circuit[:6], tau=tau)
            # only rank 0's blinders count: the others pass different ones and must end with the same proof
            mine = blinders if rank == 0 else [b + 1 + rank for b in blinders]
            proof = prv.prove(ipk, circuit[6], circuit[7], blinders=mine)
            _check_frozen_proof(proof, gp)
            assert sh.exchanges == 1 + 3 + 1, sh.exchanges        # blinders, three rounds of commitments, the openings
        else:
            from kzg_snark_amd import plonk_device
            idx = plonk_device.DeviceIndexer(curve)
            ipk, ivk = idx.preprocess(*circuit[:6], tau=tau)
            for deal in (False, True):
                sh = ProofSharding(deal_transforms=deal)
                prv = plonk_device.DeviceProver(curve, alg=idx.alg, sharding=sh)
                mine = blinders if rank == 0 else [b + 1 + rank for b in blinders]
                proof = prv.prove(ipk, circuit[6], circuit[7], blinders=mine)
                _check_frozen_proof(proof, gp)
                assert sh.exchanges == 5 + (8 if deal else 0), (deal, sh.exchanges)
                assert plonk.Verifier(curve).verify(ivk, circuit[6], proof)
            # a larger synthetic circuit: all ranks end with the same proof, and the verifier accepts it
            from kzg_snark_amd.field import GF
            F = GF(O.BLS12_381.r)
            big = plonk.synthetic_circuit(1 << 12, F, seed=12)
            idx2 = plonk_device.DeviceIndexer("bls12_381")
            ipk2, ivk2 = idx2.preprocess(*big[:6], tau=0x1234567)
            sh = ProofSharding()
            prv2 = plonk_device.DeviceProver("bls12_381", alg=idx2.alg, sharding=sh)
            proof2 = prv2.prove(ipk2, big[6], big[7])                          # random blinders: rank 0's are shared
            assert plonk.Verifier("bls12_381").verify(ivk2, big[6], proof2)
            import hashlib
            digest = hashlib.sha256(repr(sorted((k, tuple(int(c) for c in v)) for k, v in
                                                list(proof2["commitments"].items())
                                                + list(proof2["kzg_proofs"].items()))).encode()).digest()
            from kzg_snark_amd.sharding import all_gather_bytes
            assert len(set(all_gather_bytes(digest))) == 1, "ranks ended with different proofs"
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run_plonk_ranks(world, device, timeout):
    res = _run_ranks(_plonk_worker, world, timeout, (device,))
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def _sharded_ipk(ipk, OB):
    """proving key of the host indexer -> the tensors the vector-sharded prover slices (what DeviceIndexer leaves on
    the device: coefficient vectors of the eight preprocessed polynomials, sigma* values)"""
    n = ipk["subgroups"]["n"]
    co = {k: OB.tensor_of(([int(c) for c in p.list()] + [0] * n)[:n]) for k, p in ipk["polynomials"].items()}
    ss = ipk["sigma_star"]
    sv = {f"S_sigma{b + 1}": OB.tensor_of([int(v) for v in ss[b * n:(b + 1) * n]]) for b in range(3)}
    return {"ck": ipk["ck"], "coeffs": co, "sigma_values": sv, "subgroups": ipk["subgroups"]}


def _proof_key(proof):
    return (sorted((k, tuple(int(c) for c in v)) for k, v in list(proof["commitments"].items())
                   + list(proof["kzg_proofs"].items())),
            sorted((k, int(v)) for k, v in proof["evaluations"].items()))


def _sharded_worker(rank, world, port, q, case):
    """One rank of a PLONK proof whose VECTORS are split over `world` ranks (plonk_sharded.ShardedProver), the oracle
    standing in for the engine (tests/oracle_backends.py): distributed INTTs and coset NTTs through the all-to-alls,
    accumulator carries, quotient re-partition, range-sharded commitments and openings."""
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=_PG_TIMEOUT)
    try:
        OB = _oracle_backends()
        import test_plonk as TP
        from kzg_snark_amd import plonk, plonk_sharded
        from kzg_snark_amd.field import GF, PolynomialRing
        plonk.fft_ff_interpolation = lambda values, g, F: PolynomialRing(F, "X")(
            O.fft_ff_interpolation([int(v) for v in values], int(g), F.p))
        if case == "frozen":
            gp = _golden("plonk_proof_n16.json")
            circuit, curve, tau = TP.fixture_instance(), gp["curve"], int(gp["tau"], 16)
            blinders = [int(v, 16) for v in gp["blinders"]]
        else:
            curve, tau = "bn254", 0x7a75
            circuit = plonk.synthetic_circuit(64, GF(O.curve(curve).r), seed=3)
            blinders = list(range(101, 112))
        idx = plonk.Indexer(curve)
        idx.kzg = TP.oracle_backed(curve)
        ipk, ivk = idx.preprocess(*circuit[:6], tau=tau)
        sp = plonk_sharded.ShardedProver(curve, OB.OracleShardBackend(curve))
        mine = blinders if rank == 0 else [b + 1 + rank for b in blinders]       # only rank 0's count
        proof = sp.prove(_sharded_ipk(ipk, OB), circuit[6], circuit[7], blinders=mine)
        if case == "frozen":
            _check_frozen_proof(proof, gp)
        else:           # the unsharded host prover with the same blinders: the same proof, and the verifier accepts it
            ref = plonk.Prover(curve)
            ref.kzg = TP.oracle_backed(curve)
            want = ref.prove(ipk, circuit[6], circuit[7], blinders=blinders)
            assert _proof_key(proof) == _proof_key(want)
            if rank == 0:
                ver = plonk.Verifier(curve)
                assert ver.verify(ivk, circuit[6], proof)
        assert sp.tf.exchanges > 0 and sp.exchanges > 0
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "frozen"), (2, "synthetic"), (4, "synthetic")])
def test_vector_sharded_proof_over_ranks(world, case):
    """BASELINE config 5 with the vector work split over the ranks (plonk/prover.py:83-85, 243-264, 297-316 on range /
    transposed shards; commitments and openings against key shards): two ranks reproduce
    tests/golden/plonk_proof_n16.json bit for bit (n = 16 leaves a rank the minimum of 8 rows); two and four ranks
    reproduce the unsharded host prover's proof of a 64-gate circuit, which the verifier accepts."""
    res = _run_ranks(_sharded_worker, world, 600, (case,))
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_vector_sharding_needs_a_power_of_two_world():
    """Three ranks cannot split the radix-2 transforms: make_prover then keeps the dealt-MSM prover (the world-size-3
    test below), and hands out the vector-sharded one for 1, 2, 4, 8 ranks."""
    from types import SimpleNamespace
    from kzg_snark_amd import plonk_sharded
    alg = SimpleNamespace(ctx=None, r=O.BN254.r)
    for world, want in ((1, "ShardedProver"), (2, "ShardedProver"), (3, "DeviceProver"), (4, "ShardedProver"),
                        (6, "DeviceProver"), (8, "ShardedProver")):
        sh = SimpleNamespace(world=world, shard_vectors=True, group=None)
        assert type(plonk_sharded.make_prover("bn254", alg, sh)).__name__ == want, world
    sh = SimpleNamespace(world=4, shard_vectors=False, group=None)
    assert type(plonk_sharded.make_prover("bn254", alg, sh)).__name__ == "DeviceProver"


@pytest.mark.parametrize("world", [2, 3])
def test_plonk_proof_dealt_over_ranks_is_the_frozen_proof(world):
    """BASELINE config 5 across ranks (plonk/prover.py:89,113,136,184-185 dealt by sharding.ProofSharding): the
    reference's 16-gate instance proved by 2 and by 3 ranks -- commitments of a round round-robin, one opening per
    rank, blinders from rank 0 -- reproduces tests/golden/plonk_proof_n16.json bit for bit on every rank."""
    _run_plonk_ranks(world, device=False, timeout=300)


def _sharded_gpu_worker(rank, world, port, q):
    """The vector-sharded prover on the engine (plonk_sharded.GpuShardBackend), `world` ranks sharing the test box's
    GPU over gloo (tensors make the round trip through the host in the exchanges)."""
    import hashlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=_PG_TIMEOUT)
    try:
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        if here not in sys.path:
            sys.path.insert(0, here)
        import test_plonk as TP
        from kzg_snark_amd import plonk, plonk_device, plonk_sharded
        from kzg_snark_amd.field import GF
        from kzg_snark_amd.sharding import ProofSharding, all_gather_bytes
        # 1. the frozen 16-gate proof (every transform below the distributed sizes: gather, transform, keep a share)
        gp = _golden("plonk_proof_n16.json")
        circuit = TP.fixture_instance()
        curve, tau = gp["curve"], int(gp["tau"], 16)
        blinders = [int(v, 16) for v in gp["blinders"]]
        idx = plonk_device.DeviceIndexer(curve)
        ipk, ivk = idx.preprocess(*circuit[:6], tau=tau)
        prv = plonk_sharded.make_prover(curve, idx.alg, ProofSharding(shard_vectors=True))
        assert type(prv).__name__ == "ShardedProver"
        mine = blinders if rank == 0 else [b + 1 + rank for b in blinders]
        proof = prv.prove(ipk, circuit[6], circuit[7], blinders=mine)
        _check_frozen_proof(proof, gp)
        assert plonk.Verifier(curve).verify(ivk, circuit[6], proof)
        # 2. 2^12 and 2^13 gates on BLS12-381: the 4n-point (and at 2^13 the n-point) transforms go through the
        # all-to-alls and the device's column / row passes; same blinders as the unsharded device prover -> the same
        # proof on every rank, and the verifier accepts it
        F = GF(O.BLS12_381.r)
        for log_gates in (12, 13):
            big = plonk.synthetic_circuit(1 << log_gates, F, seed=log_gates)
            idx2 = plonk_device.DeviceIndexer("bls12_381")
            ipk2, ivk2 = idx2.preprocess(*big[:6], tau=0x1234567)
            bl = list(range(7, 18))
            want = plonk_device.DeviceProver("bls12_381", alg=idx2.alg).prove(ipk2, big[6], big[7], blinders=bl)
            sp = plonk_sharded.make_prover("bls12_381", idx2.alg, ProofSharding(shard_vectors=True))
            got = sp.prove(ipk2, big[6], big[7], blinders=bl if rank == 0 else [1] * 11)
            assert _proof_key(got) == _proof_key(want), log_gates
            assert plonk.Verifier("bls12_381").verify(ivk2, big[6], got)
            digest = hashlib.sha256(repr(_proof_key(got)).encode()).digest()
            assert len(set(all_gather_bytes(digest))) == 1, "ranks ended with different proofs"
            # per proof 3 + 2 (round 1, batched), 3 + 2 + 1 (round 2), 4 + 1 (round 3); + 2 once for the circuit cosets
            assert 0 < sp.tf.exchanges <= 18
            del ipk2, sp, idx2
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_vector_sharded_device_prover_on_one_gpu(world):
    """plonk_sharded.ShardedProver on the engine: one rank (every exchange degenerate, the shard logic alone) and two
    ranks sharing the GPU over gloo.  The frozen 16-gate proof bit for bit; 2^12- and 2^13-gate proofs identical to
    the unsharded device prover's for the same blinders, accepted by the verifier, the same on every rank."""
    res = _run_ranks(_sharded_gpu_worker, world, 900)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


@pytest.mark.gpu
def test_device_prover_dealt_over_two_ranks_on_one_gpu():
    """The same with the device prover (kzg_snark_amd/plonk_device.py), two ranks sharing the test box's GPU over
    gloo: the frozen 16-gate proof bit for bit with replicated and with dealt transforms, and a 2^12-gate
    BLS12-381 proof that every rank ends with identically and the host verifier accepts."""
    _run_plonk_ranks(2, device=True, timeout=600)
