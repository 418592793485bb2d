"""GPU parity of the NTT path (csrc/ntt.hip through the C ABI and the fft_ff
facade) against the oracle's restatement of fft_ff.py.  Bit-exact: integer work."""
import random

import numpy as np
import pytest

from oracle import py_oracle as O

pytestmark = pytest.mark.gpu

CURVES = ["bls12_381", "bn254"]


def _rand(r, n, rng):
    return [rng.randrange(r) for _ in range(n)]


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("log_n", [1, 2, 3, 4, 5, 6, 7, 10, 12, 13, 14])
def test_forward_matches_oracle(native, curve, log_n):
    cv = O.curve(curve)
    rng = random.Random(1000 + log_n)
    n = 1 << log_n
    w = cv.root_of_unity(n)
    x = _rand(cv.r, n, rng)
    x[0] = cv.r - 1
    if n > 2:
        x[1] = 0
    data = native.ints_to_limbs(x)
    native.get_context(curve).ntt(data, log_n, native.int_to_words(w), False)
    assert native.limbs_to_ints(data) == O.fft_ff(x, w, cv.r)


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("log_n", [1, 3, 8, 12, 13])
def test_inverse_matches_oracle(native, curve, log_n):
    cv = O.curve(curve)
    rng = random.Random(2000 + log_n)
    n = 1 << log_n
    w = cv.root_of_unity(n)
    x = _rand(cv.r, n, rng)
    data = native.ints_to_limbs(x)
    native.get_context(curve).ntt(data, log_n, native.int_to_words(w), True)
    assert native.limbs_to_ints(data) == O.ifft_ff(x, w, cv.r)


@pytest.mark.parametrize("log_n", [2, 5, 12, 13])
def test_non_primitive_w_follows_the_recursion(native, log_n):
    """fft_ff never checks w (fft_ff.py:3-37): for a w that is not a primitive n-th
    root the reference returns what its recursion computes, and so must we."""
    cv = O.BLS12_381
    rng = random.Random(3000 + log_n)
    n = 1 << log_n
    x = _rand(cv.r, n, rng)
    for w in (rng.randrange(2, cv.r), 1, cv.root_of_unity(2 * n), 0):
        data = native.ints_to_limbs(x)
        native.get_context("bls12_381").ntt(data, log_n, native.int_to_words(w), False)
        assert native.limbs_to_ints(data) == O.fft_ff(x, w, cv.r), f"w={w}"


def test_edge_values(native):
    cv = O.BLS12_381
    n, log_n = 64, 6
    w = cv.root_of_unity(n)
    ctx = native.get_context("bls12_381")
    for x in ([0] * n, [cv.r - 1] * n, [1] + [0] * (n - 1), [0] * (n - 1) + [cv.r - 1]):
        data = native.ints_to_limbs(x)
        ctx.ntt(data, log_n, native.int_to_words(w), False)
        assert native.limbs_to_ints(data) == O.fft_ff(x, w, cv.r)


@pytest.mark.parametrize("curve", CURVES)
def test_full_size_2_20_properties(native, curve):
    """BASELINE config 2: degree-2^20 NTT/INTT.  The Python oracle is too slow at
    this size, so parity is carried by (a) round trip INTT(NTT(x)) == x,
    (b) linearity on a second vector, (c) X[0] = sum x, X[n/2] = alternating sum,
    (d) a strided spot check of 16 outputs against the DFT definition."""
    cv = O.curve(curve)
    r = cv.r
    log_n = 20
    n = 1 << log_n
    w = cv.root_of_unity(n)
    rs = np.random.RandomState(7)
    raw = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64)
    raw[:, 3] >>= np.uint64(3)        # < 2^253 < r for both curves
    x = native.limbs_to_ints(raw)
    ctx = native.get_context(curve)
    ww = native.int_to_words(w)
    X = raw.copy()
    ctx.ntt(X, log_n, ww, False)
    Xi = native.limbs_to_ints(X)
    assert Xi[0] == sum(x) % r
    assert Xi[n // 2] == (sum(x[0::2]) - sum(x[1::2])) % r
    for k in (1, 2, 3, 1023, 1024, 1025, n - 1, 777777):
        wk = pow(w, k, r)
        acc, p = 0, 1
        for v in x:
            acc += v * p
            p = p * wk % r
        assert Xi[k] == acc % r, k
    back = X.copy()
    ctx.ntt(back, log_n, ww, True)
    assert np.array_equal(back, raw)
    # linearity: NTT(x + 3y) == NTT(x) + 3 NTT(y)
    y = [(v * 5 + 11) % r for v in x[:n]]
    Y = native.ints_to_limbs(y)
    ctx.ntt(Y, log_n, ww, False)
    Yi = native.limbs_to_ints(Y)
    z = [(a + 3 * b) % r for a, b in zip(x, y)]
    Z = native.ints_to_limbs(z)
    ctx.ntt(Z, log_n, ww, False)
    Zi = native.limbs_to_ints(Z)
    assert all(Zi[i] == (Xi[i] + 3 * Yi[i]) % r for i in range(0, n, 97))


def test_facade_signatures(native):
    """fft_ff / ifft_ff / fft_ff_interpolation keep the reference's call shapes."""
    from kzg_snark_amd.fft_ff import fft_ff, ifft_ff, fft_ff_interpolation
    from kzg_snark_amd.field import GF
    cv = O.BLS12_381
    F = GF(cv.r)
    g = F.root_of_unity(16)
    assert int(g) == cv.root_of_unity(16)
    vals = [F(i * i + 1) for i in range(16)]
    ev = fft_ff(vals, g, F)
    assert [int(v) for v in ev] == O.fft_ff([int(v) for v in vals], int(g), cv.r)
    co = ifft_ff(ev, g, F)
    assert co == vals
    single = [F(5)]
    assert fft_ff(single, g, F) is single          # fft_ff.py:16-17
    poly = fft_ff_interpolation([F(3)] * 16, g, F)
    assert poly.degree() == 0 and poly.list() == [F(3)]   # trailing zeros dropped (fft_ff.py:85)
    for i in range(16):
        p2 = fft_ff_interpolation(vals, g, F)
        assert p2(g ** i) == vals[i]
    arr = native.ints_to_limbs([int(v) for v in vals])       # buffer fast path: limbs in, limbs out
    out = fft_ff(arr, g, F)
    assert isinstance(out, np.ndarray) and native.limbs_to_ints(out) == [int(v) for v in ev]
    assert np.array_equal(ifft_ff(out, g, F), arr)
    with pytest.raises(AssertionError):
        fft_ff_interpolation(vals[:12], g, F)              # fft_ff.py:74
    with pytest.raises(AssertionError):
        fft_ff_interpolation(vals, F.root_of_unity(8), F)  # fft_ff.py:78


@pytest.mark.parametrize("curve,log_n,world", [("bls12_381", 14, 2), ("bls12_381", 16, 4), ("bn254", 13, 2)])
def test_distributed_ntt_rehearsal_on_one_gpu(native, curve, log_n, world):
    """kzg_ntt_columns_device / kzg_ntt_rows_exchange_device under kzg_snark_amd.sharding.DistributedNTT:
    `world` Python threads play the ranks (one engine context each, all on cuda:0) and exchange
    through an in-process all-to-all.  The result must equal the single-GPU transform and, for a
    slice, the oracle -- forward, inverse and with a w that is not a primitive root; in natural
    order (three exchanges) and in the transposed layout (two).  The coefficients of the inverse
    transform are then committed where they lie, against key shards generated in the same order
    (kzg_srs_generate_strided): the partial points must add up to p(tau) G1."""
    import threading
    import torch
    from kzg_snark_amd.kzg import KZG
    from kzg_snark_amd.sharding import DistributedNTT, GpuNttOps, transposed_index
    kzg = KZG(curve)
    tau = 0x7a75 * 0x10001 + log_n
    k1 = (log_n + 1) // 2
    N1, N2 = 1 << k1, 1 << (log_n - k1)
    cv = O.curve(curve)
    n = 1 << log_n
    rs = np.random.RandomState(log_n)
    raw = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64)
    raw[:, 3] >>= np.uint64(4)
    x = native.limbs_to_ints(raw)
    cases = [(cv.root_of_unity(n), False), (cv.root_of_unity(n), True), (0x1234567 % cv.r, False)]
    for w, inverse in cases:
        ww = native.int_to_words(w)
        ref = raw.copy()
        native.get_context(curve).ntt(ref, log_n, ww, inverse)          # single-GPU result
        if log_n <= 14:
            want = O.ifft_ff(x, w, cv.r) if inverse else O.fft_ff(x, w, cv.r)
            assert native.limbs_to_ints(ref) == want
        barrier = threading.Barrier(world)
        mailbox = [None] * world
        outs = [None] * world
        outs_t = [None] * world
        parts = [None] * world
        errs = []

        def run(rank):
            try:
                ctx = native.Context(curve)
                stream = torch.cuda.Stream(device="cuda:0")
                ctx.set_stream(stream.cuda_stream)

                def exchange(send):
                    stream.synchronize()
                    mailbox[rank] = send
                    barrier.wait()
                    recv = torch.stack([mailbox[h][rank] for h in range(world)])
                    barrier.wait()
                    return recv

                with torch.cuda.stream(stream):
                    lo, hi = rank * n // world, (rank + 1) * n // world
                    xl = torch.from_numpy(raw[lo:hi].view(np.int64)).to("cuda:0")
                    d = DistributedNTT(GpuNttOps(ctx, log_n, ww, inverse), exchange=exchange)
                    out = d.transform(xl.clone(), log_n, world=world, rank=rank)
                    stream.synchronize()
                    ctx.synchronize()
                    outs[rank] = out.cpu().numpy().view(np.uint64)
                    out_t = d.transform(xl.clone(), log_n, world=world, rank=rank, layout="transposed")
                    stream.synchronize()
                    ctx.synchronize()
                    outs_t[rank] = out_t.cpu().numpy().view(np.uint64)
                    if inverse:          # commit the shard where it lies
                        m = out_t.shape[0]
                        shard = ctx.srs_generate_strided(native.int_to_words(tau), rank * (N1 // world), m, N2, N1, 1)
                        parts[rank] = ctx.commit_device(shard, out_t.data_ptr(), [m], m)
                        shard.close()
                ctx.close()
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))
                barrier.abort()

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errs, errs
        got = np.concatenate(outs)
        assert np.array_equal(got, ref), (w == cases[2][0], inverse)
        m = n // world
        for rank in range(world):
            idx = np.array([transposed_index(log_n, world, rank, i) for i in range(m)])
            assert np.array_equal(outs_t[rank], ref[idx]), ("transposed layout", rank, inverse)
        if inverse:
            L = native.get_context(curve).fp_limbs
            acc = kzg.Z1
            for xy, inf in parts:
                assert inf[0] == 0
                v = native.limbs_to_ints(xy.reshape(2, L))
                acc = kzg.add(acc, (v[0], v[1], 1))
            p_tau = native.get_context(curve).poly_eval(n, torch.from_numpy(ref.view(np.int64)).to("cuda:0").data_ptr(), tau)
            want = kzg._g1.normalize(kzg.multiply(kzg.G1, p_tau))
            got_pt = kzg._g1.normalize(acc)
            assert (int(got_pt[0]), int(got_pt[1])) == (int(want[0]), int(want[1])), "commit of the transposed shards"


def test_full_size_2_24_properties(native):
    """The transform size of BASELINE config 4 (degree 2^24, 512 MiB per vector), device resident:
    (a) round trip, (b) DFT rows X[k] = sum_j x_j w^(jk) = x(w^k) for scattered k, evaluated by the
    device's Horner primitive (kzg_fr_poly_eval, itself checked against Python ints in
    test_vec_gpu.py), (c) linearity with the device's linear-combination primitive."""
    import torch
    cv = O.BLS12_381
    r = cv.r
    log_n = 24
    n = 1 << log_n
    w = cv.root_of_unity(n)
    ww = native.int_to_words(w)
    ctx = native.get_context("bls12_381")
    ctx.bind_torch_stream()
    g = torch.Generator(device="cuda:0").manual_seed(24)
    x = torch.randint(0, 1 << 62, (n, 4), generator=g, dtype=torch.int64, device="cuda:0")
    x[:, 3] >>= 3
    X = x.clone()
    ctx.ntt_device(X.data_ptr(), log_n, ww, False, 1)
    ctx.synchronize()
    Xh = {k: native.limbs_to_ints(X[k:k + 1].cpu().numpy().view(np.uint64))[0]
          for k in (0, 1, 2, 4095, 4096, 4097, n // 2, n - 1, 12345678)}
    for k, got in Xh.items():
        assert got == ctx.poly_eval(n, x.data_ptr(), pow(w, k, r)), k
    back = X.clone()
    ctx.ntt_device(back.data_ptr(), log_n, ww, True, 1)
    ctx.synchronize()
    assert torch.equal(back, x)
    y = torch.randint(0, 1 << 62, (n, 4), generator=g, dtype=torch.int64, device="cuda:0")
    y[:, 3] >>= 3
    z = torch.empty_like(x)
    ctx.vec_lincomb(n, [x.data_ptr(), y.data_ptr()], [n, n], [1, 3], z.data_ptr())       # x + 3y
    Y = y.clone()
    ctx.ntt_device(Y.data_ptr(), log_n, ww, False, 1)
    ctx.ntt_device(z.data_ptr(), log_n, ww, False, 1)
    want = torch.empty_like(x)
    ctx.vec_lincomb(n, [X.data_ptr(), Y.data_ptr()], [n, n], [1, 3], want.data_ptr())
    ctx.synchronize()
    assert torch.equal(z, want)


@pytest.mark.parametrize("curve,n", [("bls12_381", (1 << 16) + 1), ("bn254", 3 * (1 << 14)), ("bls12_381", (1 << 20) - 1)])
def test_ragged_lengths_at_size(native, curve, n):
    """fft_ff / ifft_ff on large lengths that are not powers of two (the reference recursion's
    result, fft_ff.py:15-37) against the C restatement of that recursion (oracle/kzg_oracle.c);
    the small cases are frozen in tests/golden/ragged_fft_vectors.json."""
    from oracle import c_oracle as CO
    cv = O.curve(curve)
    rs = np.random.RandomState(n % 1000)
    raw = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64)
    raw[:, 3] >>= np.uint64(3)
    w = 0x123456789abcdef % cv.r
    ctx = native.get_context(curve)
    for inverse in (False, True):
        got = raw.copy()
        ctx.fft_ff_any(got, native.int_to_words(w), inverse)
        want = raw.copy()
        CO.fft(curve, want, w, inverse=inverse)
        assert np.array_equal(got, want), (n, inverse)


def _uniform_below_r(rs, n, r):
    """uint64[n,4] limbs of values spread over the WHOLE range [0, r): random 256-bit words with the top limb
    folded below r's top limb (+1 where the lower limbs allow it), so near-r values occur -- the at-size inputs
    of the older tests stop at 2^253."""
    raw = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2) \
        + rs.randint(0, 2, size=(n, 4)).astype(np.uint64)
    top = r >> 192
    raw[:, 3] %= np.uint64(top)                      # value < top * 2^192 <= r
    return raw


def _plant_edge_values(raw, r, native):
    n = raw.shape[0]
    edge = [r - 1, r - 2, 0, 1, r - (1 << 20), (r - 1) // 2, (r + 1) // 2, r - 1]
    pos = [0, 1, 2, 3, n // 2, n // 2 + 1, n - 2, n - 1]
    raw[pos] = native.ints_to_limbs(edge)
    return raw


@pytest.mark.parametrize("curve,log_n", [("bls12_381", 20), ("bn254", 20), ("bls12_381", 24)])
def test_elementwise_against_the_c_oracle_at_size(native, curve, log_n):
    """BASELINE config 2 ("bit-exact vs fft_ff.py" at 2^20) and the config-4 size: every output element of the
    forward and of the inverse transform against oracle/kzg_oracle.c, which runs the recursion of
    fft_ff.py:15-37 (and :51-58) as written.  Inputs cover the whole range [0, r) with r-1, r-2, 0, 1, r-2^20
    planted, so near-r limbs go through all lazy levels of both passes and the factor-table twist."""
    from oracle import c_oracle as CO
    cv = O.curve(curve)
    n = 1 << log_n
    w = cv.root_of_unity(n)
    raw = _plant_edge_values(_uniform_below_r(np.random.RandomState(100 + log_n), n, cv.r), cv.r, native)
    ctx = native.get_context(curve)
    for inverse in (False, True):
        got = raw.copy()
        ctx.ntt(got, log_n, native.int_to_words(w), inverse)
        want = raw.copy()
        CO.fft(curve, want, w, inverse=inverse)
        assert np.array_equal(got, want), (curve, log_n, inverse)


@pytest.mark.parametrize("log_n,cases", [(12, "all"), (20, "all"), (24, "two")])
def test_worst_case_values_through_the_lazy_levels(native, log_n, cases):
    """The lazy butterflies let values grow by 4p per level (ntt.hip header: <= 49p after the 12 levels of a
    single pass).  Vectors that maximise that growth -- all r-1, alternating 0 / r-1 -- with a primitive root and
    with w = r-1 (order 2: every twiddle is +-1), at the single-pass limit 2^12, at 2^20 and at 2^24, element by
    element against the C restatement of fft_ff.py:15-37."""
    from oracle import c_oracle as CO
    cv = O.BLS12_381
    r = cv.r
    n = 1 << log_n
    top = native.ints_to_limbs([r - 1])[0]
    vec_all = np.tile(top, (n, 1))
    vec_alt = np.zeros((n, 4), dtype=np.uint64)
    vec_alt[1::2] = top
    combos = [(vec_all, cv.root_of_unity(n), False), (vec_alt, cv.root_of_unity(n), False),
              (vec_all, r - 1, False), (vec_alt, r - 1, True)]
    if cases == "two":
        combos = [combos[1], combos[2]]
    ctx = native.get_context("bls12_381")
    for vec, w, inverse in combos:
        got = vec.copy()
        ctx.ntt(got, log_n, native.int_to_words(w), inverse)
        want = vec.copy()
        CO.fft("bls12_381", want, w, inverse=inverse)
        assert np.array_equal(got, want), (log_n, hex(w)[:12], inverse)


@pytest.mark.parametrize("curve", CURVES)
def test_transform_beside_a_commit_in_flight(native, curve):
    """While an accumulate kernel of the commit pipeline is queued or running the NTT takes 1024-element tiles that fit
    beside its workgroups (ntt.hip, tile_log_pref); alone it takes 2048-element ones.  Same results either way, and
    both equal the C restatement of fft_ff.py:15-37 (forward and inverse, 2^20; 2^13 for the odd split)."""
    import torch
    from oracle import c_oracle as CO
    cv = O.curve(curve)
    ctx = native.get_context(curve)
    n_key = 1 << 20
    srs = ctx.srs_generate(native.int_to_words(0x4242424242 % cv.r), n_key)
    rs = np.random.RandomState(77)
    sc = rs.randint(0, 1 << 62, size=(4, n_key, 4)).astype(np.uint64)
    sc[:, :, 3] >>= np.uint64(3)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    L = ctx.fp_limbs
    for log_n in (20, 13):
        n = 1 << log_n
        w = cv.root_of_unity(n)
        ww = native.int_to_words(w)
        raw = _plant_edge_values(_uniform_below_r(rs, n, cv.r), cv.r, native)
        for inverse in (False, True):
            want = raw.copy()
            CO.fft(curve, want, w, inverse=inverse)
            alone = torch.from_numpy(raw.view(np.int64)).to("cuda:0")
            busy = alone.clone()
            torch.cuda.synchronize()
            ctx.ntt_device(alone.data_ptr(), log_n, ww, inverse, 1)                  # nothing in flight
            ctx.synchronize()
            assert ctx.prof_read("ntt_tile_log")[0] == 11
            xy, inf = np.zeros((4, 2 * L), dtype=np.uint64), np.zeros(4, dtype=np.uint8)
            ctx.commit_device_async(srs, d_sc.data_ptr(), [n_key] * 4, n_key, xy, inf)   # four MSMs queued
            ctx.ntt_device(busy.data_ptr(), log_n, ww, inverse, 1)                   # beside them
            took = ctx.prof_read("ntt_tile_log")[0]          # 10 unless the four MSMs had already left the queue
            ctx.commit_flush()
            ctx.synchronize()
            assert np.array_equal(alone.cpu().numpy().view(np.uint64), want), (log_n, inverse, "alone")
            assert np.array_equal(busy.cpu().numpy().view(np.uint64), want), (log_n, inverse, "beside a commit")
            assert int(inf.sum()) == 0 and took in (10, 11)
            # the choice above rides on an event query; both tiles (and the 4096-element one of the experiments)
            # deterministically through the override, so that neither path can regress unseen
            for tile_log in (10, 11, 12):
                ctx.set_tuning("ntt_tile_log", tile_log)
                try:
                    forced = torch.from_numpy(raw.view(np.int64)).to("cuda:0")
                    torch.cuda.synchronize()
                    ctx.ntt_device(forced.data_ptr(), log_n, ww, inverse, 1)
                    ctx.synchronize()
                    assert ctx.prof_read("ntt_tile_log")[0] == tile_log
                finally:
                    ctx.set_tuning("ntt_tile_log", 0)
                assert np.array_equal(forced.cpu().numpy().view(np.uint64), want), (log_n, inverse, tile_log)
    srs.close()


@pytest.mark.parametrize("curve", CURVES)
def test_transposed_to_natural_passes(native, curve):
    """kzg_ntt_rows_twist_device + kzg_ntt_columns_plain_device: the four-step taken the other way round -- input in
    the transposed layout (row rho holds the elements b N1 + rho), output in natural order -- equals the whole
    transform (fft_ff.py:15-37 / :39-58) for a primitive root, forward and inverse, 2^13 (odd split) and 2^16; on a
    row range with its base; a w that is not a primitive root is refused."""
    import torch
    cv = O.curve(curve)
    ctx = native.get_context(curve)
    rs = np.random.RandomState(5)
    for log_n in (13, 16):
        n = 1 << log_n
        k1 = (log_n + 1) // 2
        N1, N2 = 1 << k1, 1 << (log_n - k1)
        w = cv.root_of_unity(n)
        ww = native.int_to_words(w)
        raw = _plant_edge_values(_uniform_below_r(rs, n, cv.r), cv.r, native)
        for inverse in (False, True):
            want = raw.copy()
            ctx.ntt(want, log_n, ww, inverse)                                           # the whole transform
            x = torch.from_numpy(raw.view(np.int64)).to("cuda:0")
            T = x.view(N2, N1, 4).permute(1, 0, 2).contiguous()                         # T[rho][b] = x[b N1 + rho]
            torch.cuda.synchronize()             # torch's stream wrote T; the context runs on a stream of its own
            half = N1 // 2
            ctx.ntt_rows_twist_device(T.data_ptr(), log_n, ww, inverse, half, 0)        # two row ranges, as two ranks would
            ctx.ntt_rows_twist_device(T.data_ptr() + half * N2 * 32, log_n, ww, inverse, half, half)
            M = T.view(N1, N2, 4)
            ctx.ntt_columns_plain_device(M.data_ptr(), log_n, ww, inverse, N2)
            ctx.synchronize()
            assert np.array_equal(M.cpu().numpy().view(np.uint64).reshape(n, 4), want), (log_n, inverse)
    with pytest.raises(native.NativeError):
        t = torch.zeros((1 << 13, 4), dtype=torch.int64, device="cuda:0")
        ctx.ntt_rows_twist_device(t.data_ptr(), 13, native.int_to_words(12345), False, 128, 0)
