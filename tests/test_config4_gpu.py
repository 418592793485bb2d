"""BASELINE config 4 -- degree-2^24 KZG commit + batched open, polynomial sharded by coefficient
range over 8 GPUs (kzg.py:80-159 at SURVEY.md section 8a's cfg4 sizes) -- as far as ONE GPU can
test it, every result against the trapdoor identities:

  * the whole 2^24 job on one MI355X: 2^24-point key (27 GiB window table), commit of one
    polynomial, open of k = 6 polynomials (the batch plonk/prover.py:184 opens);
  * the 8-way range-sharded job, all eight ranks run one after the other on this GPU exactly as
    kzg_snark_amd/sharding.py drives them (key shards from kzg_srs_generate_range, per-rank commit,
    kzg_open_shard_begin -> carry exchange -> kzg_open_shard_finish, partial points added on the
    host): must equal the unsharded result bit for bit.

What this cannot show is the time of the exchange over xGMI; that is bench.py --gpus N."""
import numpy as np
import pytest

from oracle import py_oracle as O

pytestmark = pytest.mark.gpu

CURVE = "bls12_381"
LOG_N = 24
TAU = 0x6b7a675f736e61726b7a675f736e6172
Z = 0x1111111111111111111111111111
XI = 0x2222222222222222222222


@pytest.fixture(scope="module")
def job(native):
    import torch
    from kzg_snark_amd.kzg import KZG
    kzg = KZG(CURVE)
    ctx = native.get_context(CURVE)
    ctx.bind_torch_stream()                                # torch-side slicing below is ordered with the engine
    n = 1 << LOG_N
    k = 6
    g = torch.Generator(device="cpu").manual_seed(LOG_N)
    # ragged batch: polynomial i has n - 3*i coefficients (the provers' n+2..n+6 shapes, kzg.py:93-110)
    lens = [n - 3 * i for i in range(k)]
    polys = torch.zeros((k, n, 4), dtype=torch.int64, device="cuda:0")
    for i in range(k):
        host = torch.randint(0, 1 << 62, (lens[i], 4), generator=g, dtype=torch.int64)
        host[:, 3] >>= 3                                   # < 2^251 < r: canonical
        polys[i, :lens[i]] = host.to("cuda:0")
        del host
    torch.cuda.synchronize()
    return {"kzg": kzg, "ctx": ctx, "n": n, "k": k, "lens": lens, "polys": polys, "native": native}


def _pt(native, ctx, xy):
    v = native.limbs_to_ints(np.asarray(xy).reshape(2, ctx.fp_limbs))
    return (v[0], v[1])


def _g1_times(kzg, s):
    p3 = kzg._g1.normalize(kzg.multiply(kzg.G1, s))
    return (int(p3[0]), int(p3[1]))


def _trapdoor(job):
    """p_0(tau) and the witness scalar ((P(tau) - P(z)) / (tau - z)), P = sum xi^(i+1) p_i,
    with every evaluation done by the device's Horner kernel (kzg_fr_poly_eval; itself tested
    against Python ints in tests/test_vec_gpu.py)."""
    ctx, r = job["ctx"], job["kzg"].curve_order
    pt, pz = [], []
    for i in range(job["k"]):
        ptr = job["polys"][i].data_ptr()
        pt.append(ctx.poly_eval(job["lens"][i], ptr, TAU % r))
        pz.append(ctx.poly_eval(job["lens"][i], ptr, Z % r))
    Pt = sum(pow(XI, i + 1, r) * v for i, v in enumerate(pt)) % r
    Pz = sum(pow(XI, i + 1, r) * v for i, v in enumerate(pz)) % r
    return pt[0], Pz, (Pt - Pz) * pow((TAU - Z) % r, -1, r) % r


def test_whole_job_on_one_gpu(job):
    native, ctx, kzg, n = job["native"], job["ctx"], job["kzg"], job["n"]
    w = native.int_to_words
    srs = ctx.srs_generate(w(TAU), n)
    xy, inf = ctx.commit_device(srs, job["polys"][0].data_ptr(), [n], n)
    p0_tau, Pz, wit = _trapdoor(job)
    assert inf[0] == 0 and _pt(native, ctx, xy) == _g1_times(kzg, p0_tau), "commit(ck, p) != p(tau) G1"
    oxy, oinf, ev = ctx.open(srs, job["polys"].data_ptr(), job["lens"], n, w(Z), w(XI), device=True)
    assert native.limbs_to_ints(ev.reshape(1, 4))[0] == Pz, "P(z)"
    assert oinf[0] == 0 and _pt(native, ctx, oxy) == _g1_times(kzg, wit), "open != trapdoor witness"
    job["whole"] = (_pt(native, ctx, xy), _pt(native, ctx, oxy), Pz)
    del srs


def test_eight_way_range_sharded_job(job):
    """All 8 ranks' local work at the config's real shard size (2^21 coefficients, 3.4 GiB key
    shard each), the exchanges done on the host in between -- kzg_snark_amd/sharding.py's
    open_range / commit_range with this process playing every rank."""
    from kzg_snark_amd.sharding import range_of
    native, ctx, kzg, n, k = job["native"], job["ctx"], job["kzg"], job["n"], job["k"]
    r = kzg.curve_order
    w = native.int_to_words
    G = 8
    ranges = [range_of(g, G, n) for g in range(G)]
    assert all(hi - lo == 1 << 21 for lo, hi in ranges)

    def slice_of(g):
        lo, hi = ranges[g]
        sl = job["polys"][:, lo:hi].contiguous()
        return sl, [max(0, min(hi, L) - lo) for L in job["lens"]]

    # phase 1 (every rank): value at z of the rank's combined slice
    H = []
    for g in range(G):
        sl, sl_lens = slice_of(g)
        h = ctx.open_shard_begin(sl.data_ptr(), sl_lens, sl.shape[1], w(Z), w(XI))
        H.append(native.limbs_to_ints(h.reshape(1, 4))[0])
    # the exchange: one field element per rank (sharding.DistributedCommitter.open_range)
    carries = [sum(H[g2] * pow(Z, ranges[g2][0] - ranges[g][1], r) for g2 in range(g + 1, G)) % r
               for g in range(G)]
    # phase 2 (every rank) + the range-mode commit of polynomial 0
    proof, commit_sum, ev0 = kzg.Z1, kzg.Z1, None
    for g in range(G):
        lo, hi = ranges[g]
        sl, sl_lens = slice_of(g)
        ctx.open_shard_begin(sl.data_ptr(), sl_lens, sl.shape[1], w(Z), w(XI))
        start = 0 if g == 0 else lo - 1
        oshard = ctx.srs_generate(w(TAU), hi - 1 - start, start=start)
        xy, inf, ev = ctx.open_shard_finish(oshard, w(Z), w(carries[g]), g == 0)
        del oshard
        assert inf[0] == 0
        proof = kzg.add(proof, _pt(native, ctx, xy) + (1,))
        if g == 0:
            ev0 = native.limbs_to_ints(ev.reshape(1, 4))[0]
        cshard = ctx.srs_generate(w(TAU), hi - lo, start=lo)
        cxy, cinf = ctx.commit_device(cshard, sl[0].data_ptr(), [hi - lo], hi - lo)
        del cshard
        commit_sum = kzg.add(commit_sum, _pt(native, ctx, cxy) + (1,))
    p0_tau, Pz, wit = _trapdoor(job)
    cs, pr = kzg._g1.normalize(commit_sum), kzg._g1.normalize(proof)
    assert (int(cs[0]), int(cs[1])) == _g1_times(kzg, p0_tau), "sum of shard commitments"
    assert (int(pr[0]), int(pr[1])) == _g1_times(kzg, wit), "sum of shard proofs"
    assert ev0 == Pz
    if "whole" in job:                                     # and bit-identical to the unsharded job
        assert ((int(cs[0]), int(cs[1])), (int(pr[0]), int(pr[1])), ev0) == job["whole"]
