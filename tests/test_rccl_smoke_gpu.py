"""The collectives bench.py and kzg_snark_amd/sharding.py issue under the "nccl" (RCCL) backend, exercised on the one
GPU a test box has: a one-rank RCCL process group accepts the same calls with the same dtypes (uint8 records, int64
limb blocks, float64 / int64 reductions).  This cannot show scaling -- it shows that the calls the N-GPU run makes are
ones this RCCL build takes.  The exchange logic itself is covered with two gloo ranks in tests/test_sharding_gloo.py."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_one_rank_rccl_group_takes_the_calls_the_multi_gpu_paths_make():
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    except Exception as e:   # noqa: BLE001 -- no RCCL group on this box (an environment matter): nothing to exercise
        pytest.skip(f"a one-rank RCCL process group could not be brought up here: {e!r}")
    try:
        # bench.py Env.max_over_ranks / the verification flag
        t = torch.tensor([1.25], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.25
        f = torch.tensor([1], device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        assert int(f.item()) == 1
        # sharding.all_gather_bytes: fixed-size uint8 records, gathered into ONE device tensor
        rec = torch.frombuffer(bytearray(range(97)), dtype=torch.uint8).to(dev)
        block = torch.empty(97, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(block, rec)
        assert bytes(block.cpu().numpy().tobytes()) == bytes(range(97))
        # sharding.DistributedNTT.exchange: blocks of 32-byte field elements as int64[.., 4]
        send = torch.arange(4 * 1024, dtype=torch.int64, device=dev).view(1024, 4).contiguous()
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send)
        assert torch.equal(recv, send)
        dist.barrier()
        torch.cuda.synchronize(dev)

        # the same through the product's own helpers, forced to issue their collective in this one-rank group:
        # a G1 record, and a whole 2^14 transform through DistributedNTT on the device (both layouts)
        import numpy as np
        from kzg_snark_amd import _native
        from kzg_snark_amd.sharding import DistributedNTT, GpuNttOps, all_gather_bytes, pack_point, unpack_point
        pt = (12345, 67890, 1)
        assert [unpack_point(b) for b in all_gather_bytes(pack_point(pt), always=True)] == [pt]
        # the gather keeps one device buffer per record length and reuses it: repeated exchanges of two lengths,
        # interleaved, return what was put in (and allocate nothing new)
        from kzg_snark_amd import sharding as _sh
        before = len(_sh._gather_bufs)
        for i in range(6):
            rec32, rec97 = bytes([i] * 32), bytes([(i * 7 + j) % 251 for j in range(97)])
            assert all_gather_bytes(rec32, always=True) == [rec32]
            assert all_gather_bytes(rec97, always=True) == [rec97]
        assert len(_sh._gather_bufs) == before + 1            # 97-byte records had their buffer already
        assert all_gather_bytes(b"", always=True) == [b""]
        ctx = _native.get_context("bls12_381")
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            ctx.bind_torch_stream(stream)
            log_n, r = 14, 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
            w = pow(7, (r - 1) >> log_n, r)
            ww = _native.int_to_words(w)
            rs = np.random.RandomState(4)
            host = rs.randint(0, 1 << 62, size=(1 << log_n, 4)).astype(np.uint64)
            host[:, 3] >>= np.uint64(3)
            want = host.copy()
            ctx.ntt(want, log_n, ww, False)                                   # single-GPU transform, host entry
            x = torch.from_numpy(host.view(np.int64)).to(dev)
            d = DistributedNTT(GpuNttOps(ctx, log_n, ww, False))
            d.always_exchange = True
            got = d.transform(x, log_n, layout="natural")
            ctx.synchronize()
            assert np.array_equal(got.cpu().numpy().view(np.uint64), want)
        ctx.bind_torch_stream(torch.cuda.default_stream(dev))

        # BASELINE config 5 over ranks (sharding.ProofSharding): the frozen 16-gate proof through the device prover with
        # every exchange issued as an RCCL collective in this one-rank group -- the record gathers of the rounds, the
        # blinders, and (deal_transforms) the eight broadcasts of device tensors
        import json
        from kzg_snark_amd import plonk, plonk_device, sharding
        here = os.path.dirname(os.path.abspath(__file__))
        gp = json.load(open(os.path.join(here, "golden", "plonk_proof_n16.json")))
        g16 = json.load(open(os.path.join(here, "golden", "plonk_instance_n16.json")))
        col = {k: [int(x, 16) for x in v] for k, v in g16["columns"].items()}
        w_full = col["a"] + col["b"] + col["c"]
        sharding.FORCE_COLLECTIVES = True
        try:
            idx = plonk_device.DeviceIndexer(gp["curve"])
            ipk, ivk = idx.preprocess(col["qM"], col["qL"], col["qR"], col["qO"], col["qC"], g16["perm"],
                                      tau=int(gp["tau"], 16))
            for deal in (False, True):
                sh = sharding.ProofSharding(deal_transforms=deal)
                prv = plonk_device.DeviceProver(gp["curve"], alg=idx.alg, sharding=sh)
                proof = prv.prove(ipk, w_full[:5], w_full[5:], blinders=[int(v, 16) for v in gp["blinders"]])
                for k, v in gp["proof"]["commitments"].items():
                    assert tuple(int(c) for c in proof["commitments"][k]) == (int(v[0], 16), int(v[1], 16), v[2]), k
                for k, v in gp["proof"]["kzg_proofs"].items():
                    assert tuple(int(c) for c in proof["kzg_proofs"][k]) == (int(v[0], 16), int(v[1], 16), v[2]), k
                assert plonk.Verifier(gp["curve"]).verify(ivk, w_full[:5], proof)
        finally:
            sharding.FORCE_COLLECTIVES = False
    finally:
        dist.destroy_process_group()
