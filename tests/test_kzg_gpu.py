"""GPU parity of SRS generation, KZG.commit and KZG.open (csrc/msm.hip, csrc/poly.hip
through the C ABI and the KZG facade) against the oracle's restatement of kzg.py.
Bit-exact on affine coordinates (integer work; SURVEY.md 7.2: projective
representatives are not comparable, affine points are)."""
import random

import numpy as np
import pytest

from oracle import py_oracle as O

pytestmark = pytest.mark.gpu

CURVES = ["bls12_381", "bn254"]


def aff(pt):
    """facade point -> oracle-style affine (x, y) or None."""
    return None if pt[2] == 0 else (pt[0], pt[1])


@pytest.fixture(scope="module")
def kzgs():
    from kzg_snark_amd.kzg import KZG
    return {c: KZG(c) for c in CURVES}


@pytest.fixture(scope="module")
def small_keys(kzgs):
    out = {}
    for c in CURVES:
        tau = 0x1234567890abcdef1234567890abcdef % O.curve(c).r
        ck, tau_g2 = kzgs[c].setup(63, tau=tau)
        out[c] = (ck, tau)
    return out


@pytest.mark.parametrize("curve", CURVES)
def test_setup_matches_oracle(kzgs, small_keys, curve):
    cv = O.curve(curve)
    ck, tau = small_keys[curve]
    assert len(ck) == 64
    ref = O.setup(63, tau, cv)
    for i in (0, 1, 2, 3, 31, 62, 63):
        assert aff(ck[i]) == O.normalize(ref[i], cv), i
    assert ck[0] == kzgs[curve].G1


@pytest.mark.parametrize("curve", CURVES)
def test_commit_matches_oracle(kzgs, small_keys, curve):
    cv = O.curve(curve)
    kzg = kzgs[curve]
    ck, tau = small_keys[curve]
    rng = random.Random(11)
    ref_ck = O.setup(63, tau, cv)
    polys = [
        [rng.randrange(cv.r) for _ in range(64)],
        [rng.randrange(cv.r) for _ in range(17)],
        [0, 0, 5, 0, cv.r - 1],                       # zero coefficients are skipped (kzg.py:113-114)
        [1] * 64,
        [0] * 63 + [1],
        [cv.r - 1] * 9,
        [7],
    ]
    got = kzg.commit(ck, polys)
    want = O.commit(ref_ck, polys, cv)
    for g, w in zip(got, want):
        assert aff(g) == O.normalize(w, cv)


@pytest.mark.parametrize("curve", CURVES)
def test_commit_edge_cases(kzgs, small_keys, curve):
    cv = O.curve(curve)
    kzg = kzgs[curve]
    ck, tau = small_keys[curve]
    assert kzg.commit(ck, [[]]) == [kzg.Z1]                    # zero polynomial (kzg.py:109)
    assert kzg.commit(ck, [[0, 0, 0]]) == [kzg.Z1]
    assert kzg.commit(ck, []) == []
    assert kzg.commit(ck, [[5] + [0] * 100]) == kzg.commit(ck, [[5]])    # trailing zeros are not degree
    with pytest.raises(ValueError):                            # kzg.py:103-106
        kzg.commit(ck, [[1] * 65])
    # polynomial objects (R(list)) are accepted like lists (kzg.py:93-97)
    p = kzg.R([3, 0, 9])
    assert kzg.commit(ck, [p]) == kzg.commit(ck, [[3, 0, 9]])
    # p(X) and -p(X) commit to opposite points; their sum is the zero polynomial
    a = kzg.commit(ck, [[1, 2, 3]])[0]
    b = kzg.commit(ck, [[cv.r - 1, cv.r - 2, cv.r - 3]])[0]
    assert kzg.add(a, b) == kzg.Z1


@pytest.mark.parametrize("curve", CURVES)
def test_commit_with_plain_list_key_duplicates_and_infinity(kzgs, curve):
    """`ck` given the way the reference's callers hold it (a list of point tuples),
    with duplicate points (forces P+P inside a bucket) and a point at infinity."""
    cv = O.curve(curve)
    kzg = kzgs[curve]
    g = O.from_affine(cv.g1)
    pts = [g, g, O.double(g, cv), g, O.Z1(), O.multiply(g, 5, cv), O.neg(g, cv), g]
    ck = []
    for p in pts:
        n = O.normalize(p, cv)
        ck.append((1, 1, 0) if n is None else (n[0], n[1], 1))
    rng = random.Random(5)
    polys = [[3, 3, 3, 3, 9, 1, 3, 3],            # equal digits on equal points -> doubling in a bucket
             [1, 1, 0, 0, 0, 0, 2, 0],            # G + G - 2G = O
             [rng.randrange(cv.r) for _ in range(8)]]
    got = kzg.commit(ck, polys)
    want = O.commit(pts, polys, cv)
    for g_, w in zip(got, want):
        assert aff(g_) == O.normalize(w, cv)
    assert got[1] == kzg.Z1


@pytest.mark.parametrize("curve", CURVES)
def test_open_matches_oracle(kzgs, small_keys, curve):
    cv = O.curve(curve)
    kzg = kzgs[curve]
    ck, tau = small_keys[curve]
    ref_ck = O.setup(63, tau, cv)
    rng = random.Random(21)
    for lens in ([64], [64, 40, 7], [5, 64], [1], [2, 1], [33], [64] * 7):
        polys = [[rng.randrange(cv.r) for _ in range(n)] for n in lens]
        z, xi = rng.randrange(cv.r), rng.randrange(cv.r)
        got = kzg.open(ck, polys, z, xi)
        want, _ = O.open_(ref_ck, polys, z, xi, cv)
        assert aff(got) == O.normalize(want, cv), lens
        assert aff(got) == O.normalize(O.open_trapdoor(polys, z, xi, tau, cv), cv)
    # z = 0, xi = 0, z = a root of the polynomial
    polys = [[rng.randrange(cv.r) for _ in range(10)]]
    for z, xi in ((0, 5), (7, 0), (0, 0), (cv.r - 1, 1)):
        got = kzg.open(ck, polys, z, xi)
        want, _ = O.open_(ref_ck, polys, z, xi, cv)
        assert aff(got) == O.normalize(want, cv), (z, xi)
    assert kzg.open(ck, [[4]], 3, 2) == kzg.Z1          # constant polynomial: witness is 0


def test_open_returns_evaluation(native, small_keys):
    cv = O.BLS12_381
    ck, tau = small_keys["bls12_381"]
    rng = random.Random(3)
    polys = [[rng.randrange(cv.r) for _ in range(n)] for n in (50, 20)]
    z, xi = rng.randrange(cv.r), rng.randrange(cv.r)
    arr = np.zeros((2, 50, 4), dtype=np.uint64)
    for i, p in enumerate(polys):
        arr[i, :len(p)] = native.ints_to_limbs(p)
    ctx = native.get_context("bls12_381")
    xy, inf, ev = ctx.open(ck.srs, arr, [50, 20], 50, native.int_to_words(z), native.int_to_words(xi))
    assert native.limbs_to_ints(ev.reshape(1, 4))[0] == O.poly_eval(O.combine(polys, xi, cv.r), z, cv.r)


@pytest.mark.parametrize("curve", CURVES)
def test_commit_medium_trapdoor(kzgs, curve):
    """n = 2^12 + 3 (not a power of two, like the provers' n+2..n+6): trapdoor identity."""
    cv = O.curve(curve)
    kzg = kzgs[curve]
    tau = 0xdeadbeefcafebabe0123456789 % cv.r
    n = (1 << 12) + 3
    ck, _ = kzg.setup(n - 1, tau=tau)
    rng = random.Random(77)
    p = [rng.randrange(cv.r) for _ in range(n)]
    got = kzg.commit(ck, [p])[0]
    assert aff(got) == O.normalize(O.commit_trapdoor(p, tau, cv), cv)
    small = [rng.randrange(3) for _ in range(n)]          # PLONK-like: many 0 / 1 / 2 scalars
    assert aff(kzg.commit(ck, [small])[0]) == O.normalize(O.commit_trapdoor(small, tau, cv), cv)


def test_commit_2_18_bn254_trapdoor(native, kzgs):
    """2^18 + 5 coefficients on BN254: the 20-bit-window path (keys of >= 2^18 points) on the
    second curve, with a length that is not a power of two; includes skewed scalars."""
    cv = O.BN254
    kzg = kzgs["bn254"]
    n = (1 << 18) + 5
    tau = 0x51ab3c7d9e % cv.r
    ck, _ = kzg.setup(n - 1, tau=tau)
    rs = np.random.RandomState(19)
    raw = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64)
    raw[:, 3] >>= np.uint64(4)
    raw[1000:3000] = 0
    raw[3000:5000, 1:] = 0                         # small scalars: only the lowest windows are populated
    raw[5000:9000] = raw[5000]                     # many equal scalars: a few heavy buckets
    coeffs = native.limbs_to_ints(raw)
    ctx = native.get_context("bn254")
    xy, inf = ctx.commit(ck.srs, raw.reshape(1, n, 4), [n], n)
    got = native.limbs_to_ints(xy.reshape(2, 4))
    assert inf[0] == 0 and (got[0], got[1]) == O.normalize(O.commit_trapdoor(coeffs, tau, cv), cv)


def test_ntt_and_commit_2_22(native, kzgs):
    """Beyond the benchmark size: 2^22 evaluations -> INTT -> commit against a 2^22-point key
    (device-resident pipeline), trapdoor identity on the result; NTT checked by its round trip
    and by X[0] = sum(x)."""
    import torch
    cv = O.BLS12_381
    kzg = kzgs["bls12_381"]
    ctx = native.get_context("bls12_381")
    log_n = 22
    n = 1 << log_n
    tau = 0x77aa55cc33 % cv.r
    ck, _ = kzg.setup(n - 1, tau=tau)
    rs = np.random.RandomState(22)
    raw = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64)
    raw[:, 3] >>= np.uint64(3)
    w = cv.root_of_unity(n)
    ww = native.int_to_words(w)
    d = torch.from_numpy(raw.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    ctx.ntt_device(d.data_ptr(), log_n, ww, True, 1)                 # evaluations -> coefficients
    ctx.synchronize()
    coeff_raw = d.cpu().numpy().view(np.uint64)
    coeffs = native.limbs_to_ints(coeff_raw)
    x = native.limbs_to_ints(raw)
    assert coeffs[0] == sum(x) * pow(n, -1, cv.r) % cv.r               # c_0 = mean of the evaluations
    xy, inf = ctx.commit_device(ck.srs, d.data_ptr(), [n], n)
    got = native.limbs_to_ints(xy.reshape(2, 6))
    assert inf[0] == 0 and (got[0], got[1]) == O.normalize(O.commit_trapdoor(coeffs, tau, cv), cv)
    ctx.ntt_device(d.data_ptr(), log_n, ww, False, 1)                # back to evaluations
    ctx.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), raw)


def test_commit_full_size_2_20_trapdoor(native, kzgs):
    """BASELINE config 3: degree-2^20 commit on BLS12-381 against a 2^20-point SRS.
    The oracle's naive commit is hours at this size; parity is the trapdoor
    identity commit(ck, p) == p(tau)*G1, plus spot checks
    of the generated SRS against tau^i * G1."""
    cv = O.BLS12_381
    kzg = kzgs["bls12_381"]
    n = 1 << 20
    tau = 0x6b7a675f736e61726b7a675f736e6172 % cv.r
    ck, _ = kzg.setup(n - 1, tau=tau)
    g = O.from_affine(cv.g1)
    for i in (0, 1, 2, 65537, n - 1):
        assert aff(ck[i]) == O.normalize(O.multiply(g, pow(tau, i, cv.r), cv), cv), i
    rs = np.random.RandomState(9)
    raw = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64)
    raw[:, 3] >>= np.uint64(3)
    coeffs = native.limbs_to_ints(raw)
    ctx = native.get_context("bls12_381")
    xy, inf = ctx.commit(ck.srs, raw.reshape(1, n, 4), [n], n)
    assert inf[0] == 0
    got = native.limbs_to_ints(xy.reshape(2, 6))
    want = O.normalize(O.commit_trapdoor(coeffs, tau, cv), cv)
    assert (got[0], got[1]) == want
    # open at full size: trapdoor identity for the witness
    z, xi = 0x1111111111111111111111111111 % cv.r, 0x2222222222222222222222 % cv.r
    oxy, oinf, ev = ctx.open(ck.srs, raw.reshape(1, n, 4), [n], n, native.int_to_words(z), native.int_to_words(xi))
    got = native.limbs_to_ints(oxy.reshape(2, 6))
    want = O.normalize(O.open_trapdoor([coeffs], z, xi, tau, cv), cv)
    assert (got[0], got[1]) == want
    assert native.limbs_to_ints(ev.reshape(1, 4))[0] == xi * O.poly_eval(coeffs, z, cv.r) % cv.r


@pytest.mark.parametrize("curve", CURVES)
def test_reference_selftest_recipe(kzgs, curve):
    """The reference's only test of this path, kzg.py:291-380, replayed on the engine:
    setup(5); three lists of two polynomials; commit, open, check per list, batch_check,
    batch == all(individual); then a corrupted evaluation must be rejected by both."""
    kzg = kzgs[curve]
    Fq = kzg.Fq
    ck, rk = kzg.setup(5)                                   # unseeded tau, as in the reference
    lists = [[[1, 2, 3, 4, 5], [5, 4, 3, 2, 1]],
             [[2, 0, 1], [0, 0, 0, 7]],
             [[9, 8, 7, 6, 5, 4], [1]]]
    commitments_list = [kzg.commit(ck, polys) for polys in lists]
    z_list = [Fq.random_element() for _ in lists]
    xi_list = [Fq.random_element() for _ in lists]
    evaluations_list = [[kzg.R(p)(z) for p in polys] for polys, z in zip(lists, z_list)]
    proof_list = [kzg.open(ck, polys, z, xi) for polys, z, xi in zip(lists, z_list, xi_list)]
    individual = [kzg.check(rk, c, z, e, p, xi)
                  for c, z, e, p, xi in zip(commitments_list, z_list, evaluations_list, proof_list, xi_list)]
    assert all(individual)
    assert kzg.batch_check(rk, commitments_list, z_list, evaluations_list, proof_list, xi_list)
    bad = [list(e) for e in evaluations_list]
    bad[1][0] = bad[1][0] + 1
    assert not kzg.check(rk, commitments_list[1], z_list[1], bad[1], proof_list[1], xi_list[1])
    assert not kzg.batch_check(rk, commitments_list, z_list, bad, proof_list, xi_list)


@pytest.mark.parametrize("curve", CURVES)
def test_sharded_open_and_commit_by_coefficient_range(native, kzgs, curve):
    """Range-mode sharding rehearsed on one GPU: the 'ranks' run one after the
    other; key shards come from kzg_srs_generate_range; partial points are added on the host.
    Must equal the unsharded open / commit (and the trapdoor identity)."""
    import torch
    from kzg_snark_amd.sharding import range_of
    cv = O.curve(curve)
    kzg = kzgs[curve]
    ctx = native.get_context(curve)
    rng = random.Random(31)
    n, world = 1000, 3
    tau = rng.randrange(cv.r)
    lens = [n, 700, 333]                                  # ragged: later polynomials end inside a shard
    polys = [[rng.randrange(cv.r) for _ in range(m)] for m in lens]
    z, xi = rng.randrange(cv.r), rng.randrange(cv.r)
    zw, xw, tw = native.int_to_words(z), native.int_to_words(xi), native.int_to_words(tau)
    L = ctx.fp_limbs

    def to_pt(xy, inf):
        if inf:
            return kzg.Z1
        v = native.limbs_to_ints(xy.reshape(2, L))
        return (v[0], v[1], 1)

    ranges = [range_of(g, world, n) for g in range(world)]
    # phase 1 on every rank: slice evaluation H_g
    dev_slices, H = [], []
    for lo, hi in ranges:
        arr = np.zeros((len(polys), hi - lo, 4), dtype=np.uint64)
        sl_lens = []
        for i, p in enumerate(polys):
            part = p[lo:hi]
            sl_lens.append(len(part))
            if part:
                arr[i, :len(part)] = native.ints_to_limbs(part)
        t = torch.from_numpy(arr.view(np.int64)).to("cuda:0")
        dev_slices.append((t, sl_lens))
        h = ctx.open_shard_begin(t.data_ptr(), sl_lens, hi - lo, zw, xw)
        H.append(native.limbs_to_ints(h.reshape(1, 4))[0])
    # exchange (host): carry_g = sum_{g' > g} H_g' * z^(lo_g' - hi_g)
    carries = []
    for g, (lo, hi) in enumerate(ranges):
        carries.append(sum(H[g2] * pow(z, ranges[g2][0] - hi, cv.r) for g2 in range(g + 1, world)) % cv.r)
    # phase 2 on every rank
    proof, commit_sum, ev0 = kzg.Z1, kzg.Z1, None
    for g, (lo, hi) in enumerate(ranges):
        t, sl_lens = dev_slices[g]
        ctx.open_shard_begin(t.data_ptr(), sl_lens, hi - lo, zw, xw)      # a context holds one slice at a time
        start = 0 if g == 0 else lo - 1
        shard = ctx.srs_generate(tw, hi - 1 - start, start=start)
        xy, inf, ev = ctx.open_shard_finish(shard, zw, native.int_to_words(carries[g]), g == 0)
        proof = kzg.add(proof, to_pt(xy, inf[0]))
        if g == 0:
            ev0 = native.limbs_to_ints(ev.reshape(1, 4))[0]
        # range-mode commit of the first polynomial with a shard starting at lo
        cshard = ctx.srs_generate(tw, hi - lo, start=lo)
        cxy, cinf = ctx.commit_device(cshard, t.data_ptr(), [hi - lo], hi - lo)
        commit_sum = kzg.add(commit_sum, to_pt(cxy[0], cinf[0]))
    want = O.normalize(O.open_trapdoor(polys, z, xi, tau, cv), cv)
    assert (proof[0], proof[1]) == want
    assert ev0 == O.poly_eval(O.combine(polys, xi, cv.r), z, cv.r)
    assert (commit_sum[0], commit_sum[1]) == O.normalize(O.commit_trapdoor(polys[0], tau, cv), cv)


def test_key_file_round_trip(kzgs, tmp_path):
    """On-disk SRS format: save, reload, same points, same commitments."""
    kzg = kzgs["bls12_381"]
    ck, _ = kzg.setup(300, tau=987654321)
    path = tmp_path / "srs.bin"
    kzg.save_key(ck, str(path), chunk=128)
    assert path.stat().st_size == 24 + 301 * (96 + 1)
    ck2 = kzg.load_key(str(path))
    assert len(ck2) == 301 and all(ck2[i] == ck[i] for i in (0, 1, 150, 300))
    poly = list(range(1, 302))
    assert kzg.commit(ck2, [poly]) == kzg.commit(ck, [poly])
    with pytest.raises(ValueError):
        kzgs["bn254"].load_key(str(path))


def test_skewed_scalars_2_20(native, kzgs):
    """Heavy buckets at full size: all-ones (one bucket receives 2^20 entries) and {0,1,2}
    coefficients (the PLONK-like case of SURVEY.md 8d).  Trapdoor identities."""
    cv = O.BLS12_381
    kzg = kzgs["bls12_381"]
    n = 1 << 20
    tau = 0x3141592653589793 % cv.r
    ck, _ = kzg.setup(n - 1, tau=tau)
    ctx = native.get_context("bls12_381")
    g = O.from_affine(cv.g1)
    ones = np.zeros((n, 4), dtype=np.uint64)
    ones[:, 0] = 1
    xy, inf = ctx.commit(ck.srs, ones.reshape(1, n, 4), [n], n)
    geo = (pow(tau, n, cv.r) - 1) * pow(tau - 1, -1, cv.r) % cv.r          # 1 + tau + ... + tau^(n-1)
    assert tuple(native.limbs_to_ints(xy.reshape(2, 6))) == O.normalize(O.multiply(g, geo, cv), cv)
    rs = np.random.RandomState(5)
    small = np.zeros((n, 4), dtype=np.uint64)
    small[:, 0] = rs.randint(0, 3, size=n)
    xy, inf = ctx.commit(ck.srs, small.reshape(1, n, 4), [n], n)
    coeffs = [int(v) for v in small[:, 0]]
    assert tuple(native.limbs_to_ints(xy.reshape(2, 6))) == O.normalize(O.commit_trapdoor(coeffs, tau, cv), cv)
    # one BIN of the two-step partition receives every entry, spread over all of its buckets (scalars
    # below 2^10 fill the 1024 buckets of bin 0): that bin is cut into 2^20 / 8192 chunks, each with
    # a histogram over many buckets; and the same shifted into the second window (multiples of 2^20)
    for shift in (0, 20):
        vals = rs.randint(0, 1 << 10, size=n).astype(np.uint64)
        sc = np.zeros((n, 4), dtype=np.uint64)
        sc[:, 0] = vals << np.uint64(shift)
        xy, inf = ctx.commit(ck.srs, sc.reshape(1, n, 4), [n], n)
        coeffs = [int(v) << shift for v in vals]
        assert tuple(native.limbs_to_ints(xy.reshape(2, 6))) == O.normalize(O.commit_trapdoor(coeffs, tau, cv), cv)


def test_pipelined_commit_across_calls_with_buffer_reuse(native, kzgs):
    """kzg_commit_device_async / kzg_commit_flush -- the path bench.py's headline loop times: more
    polynomials than the pipeline has slots, issued over several calls, and the SAME scalar buffer
    overwritten between the calls by work on the context's stream (legal: the call makes the
    stream wait until prep has consumed the scalars; include/kzg_mi355x.h).  Every result against
    the trapdoor identity, so a slot-recycling or stream-ordering regression cannot hide."""
    import torch
    cv = O.BLS12_381
    kzg = kzgs["bls12_381"]
    ctx = native.Context("bls12_381")
    stream = torch.cuda.Stream(device="cuda:0")
    ctx.bind_torch_stream(stream)
    n = (1 << 18) + 3                         # 20-bit windows, not a power of two
    tau = 0x5151515151515151515151 % cv.r
    srs = ctx.srs_generate(native.int_to_words(tau), n)
    lens = [n, n - 5, n // 2]
    rounds, L = 4, ctx.fp_limbs               # 12 polynomials through 4 slots
    g = torch.Generator(device="cuda:0").manual_seed(77)
    outs, want = [], []
    with torch.cuda.stream(stream):
        buf = torch.zeros((len(lens), n, 4), dtype=torch.int64, device="cuda:0")
        for _ in range(rounds):
            fresh = torch.randint(0, 1 << 62, (len(lens), n, 4), generator=g, dtype=torch.int64, device="cuda:0")
            fresh[..., 3] >>= 3
            buf.copy_(fresh)                  # overwrites the scalars of the previous call on the context's stream
            want.append([ctx.poly_eval(m, fresh[i].data_ptr(), tau) for i, m in enumerate(lens)])
            xy = np.zeros((len(lens), 2 * L), dtype=np.uint64)
            inf = np.zeros(len(lens), dtype=np.uint8)
            outs.append((xy, inf))
            ctx.commit_device_async(srs, buf.data_ptr(), lens, n, xy, inf)
        ctx.commit_flush()
    G1 = O.from_affine(cv.g1)
    for (xy, inf), evs in zip(outs, want):
        for i, p_tau in enumerate(evs):
            assert inf[i] == 0
            assert tuple(native.limbs_to_ints(xy[i].reshape(2, L))) == O.normalize(O.multiply(G1, p_tau, cv), cv)
    ctx.close()


def test_bind_torch_stream_orders_the_default_stream(native):
    """Context.bind_torch_stream() with torch's DEFAULT stream (HIP's null stream, handle 0 -- which
    kzg_ctx_set_stream would read as "back to the private stream"): a torch copy followed at once
    by an engine call must see the copied data, and the commit path must accept that stream."""
    import torch
    ctx = native.Context("bls12_381")
    default = torch.cuda.default_stream()
    assert default.cuda_stream == 0
    with torch.cuda.stream(default):             # other tests may have left another stream current
        assert ctx.bind_torch_stream().cuda_stream == 0
        n = 1 << 21
        base = torch.randint(0, 1 << 62, (2 * n, 4), dtype=torch.int64, device="cuda:0")
        base[:, 3] >>= 3
        torch.cuda.synchronize()
        for it in range(8):
            sl = base[it:it + 2 * n:2].contiguous()              # strided copy on the default stream
            got = ctx.poly_eval(n, sl.data_ptr(), 12345)          # read at once by the engine
            torch.cuda.synchronize()
            assert got == ctx.poly_eval(n, sl.data_ptr(), 12345)
        srs = ctx.srs_generate(native.int_to_words(7), 1 << 10)
        xy, inf = ctx.commit_device(srs, base.data_ptr(), [1 << 10], 1 << 10)     # events / waits on the null stream
        assert inf[0] == 0
    ctx.close()


def test_list_form_key_cache_is_bounded_and_notices_edits(kzgs):
    """KZG._key: list-form commitment keys (what the reference's callers pass on every call, kzg.py:80)
    are uploaded once and cached by id(); the cache holds at most two device tables, frees the evicted
    one, and a list edited in place is uploaded again instead of silently reusing the old table."""
    kzg = kzgs["bn254"]
    ck0, _ = kzg.setup(15, tau=11)
    lists = [[ck0[i] for i in range(16)] for _ in range(3)]
    poly = list(range(1, 17))
    want = kzg.commit(ck0, [poly])
    for lst in lists:
        assert kzg.commit(lst, [poly]) == want
    assert len(kzg._loaded) == kzg._KEY_CACHE and id(lists[0]) not in kzg._loaded
    first = kzg._loaded[id(lists[2])][0]
    assert kzg.commit(lists[2], [poly]) == want and kzg._loaded[id(lists[2])][0] is first     # cache hit
    lists[2][5] = kzg.multiply(kzg.G1, 12345)                                                   # in-place edit
    got = kzg.commit(lists[2], [poly])
    assert got != want and kzg._loaded[id(lists[2])][0] is not first
    expect = kzg.add(want[0], kzg.multiply(kzg.add(lists[2][5], kzg.neg(ck0[5])), poly[5]))
    assert got[0] == expect


@pytest.mark.parametrize("curve", CURVES)
def test_pipelined_open_matches_the_synchronous_one(native, curve):
    """kzg_open_device_async: more openings than pipeline slots, interleaved with pipelined commits, the SAME device
    buffer refilled between calls; every proof point and P(z) must equal what kzg_open_device returns for the same
    inputs (which the other tests pin to the oracle and the trapdoor), incl. a constant and an all-zero batch."""
    import torch
    cv = O.curve(curve)
    ctx = native.Context(curve)
    stream = torch.cuda.Stream(device="cuda:0")
    ctx.bind_torch_stream(stream)
    L = ctx.fp_limbs
    n = (1 << 16) + 5
    tau = 0x7777777777 % cv.r
    srs = ctx.srs_generate(native.int_to_words(tau), n)
    rng = random.Random(99)
    g = torch.Generator(device="cuda:0").manual_seed(5)
    cases = [([n, n - 1, 777], False), ([n], False), ([1, 1], False), ([0, 0], True), ([n - 7, n], False),
             ([n, n, n, n, n, n], False), ([5], False)]
    jobs = []
    with torch.cuda.stream(stream):
        buf = torch.zeros((6, n, 4), dtype=torch.int64, device="cuda:0")
        for lens, _ in cases:
            k = len(lens)
            fresh = torch.randint(0, 1 << 62, (6, n, 4), generator=g, dtype=torch.int64, device="cuda:0")
            fresh[..., 3] >>= 3
            z, xi = rng.randrange(cv.r), rng.randrange(cv.r)
            zw, xw = native.int_to_words(z), native.int_to_words(xi)
            want = ctx.open(srs, fresh.data_ptr(), lens, n, zw, xw, device=True)       # synchronous reference
            buf.copy_(fresh)                                                             # overwrite the shared buffer
            out = (np.zeros(2 * L, dtype=np.uint64), np.zeros(1, dtype=np.uint8), np.zeros(4, dtype=np.uint64))
            ctx.open_device_async(srs, buf.data_ptr(), lens[:k], n, zw, xw, *out)
            cxy, cinf = np.zeros((1, 2 * L), dtype=np.uint64), np.zeros(1, dtype=np.uint8)
            ctx.commit_device_async(srs, buf.data_ptr(), [lens[0]], n, cxy, cinf)      # shares the slots
            cwant = ctx.poly_eval(lens[0], fresh.data_ptr(), tau) if lens[0] else None
            jobs.append((want, out, (cxy, cinf, cwant)))
        ctx.commit_flush()
    G1 = O.from_affine(cv.g1)
    for (wxy, winf, wev), (xy, inf, ev), (cxy, cinf, cwant) in jobs:
        assert inf[0] == winf[0] and np.array_equal(xy, wxy) and np.array_equal(ev, wev)
        if cwant is None:
            assert cinf[0] == 1
        else:
            assert tuple(native.limbs_to_ints(cxy[0].reshape(2, L))) == O.normalize(O.multiply(G1, cwant, cv), cv)
    ctx.close()


@pytest.mark.parametrize("curve", CURVES)
def test_limb_buffers_are_accepted_like_lists(native, kzgs, small_keys, curve):
    """KZG.commit / KZG.open with uint64[n, 4] limb buffers (the fast path that skips Python-int marshalling,
    as fft_ff has it): same points as with the lists, trailing zero coefficients and the degree check included."""
    import numpy as np
    cv = O.curve(curve)
    kzg = kzgs[curve]
    ck, _ = small_keys[curve]
    rng = random.Random(77)
    polys = [[rng.randrange(cv.r) for _ in range(n)] for n in (64, 17, 1)] + [[3, 0, 5, 0, 0], [0, 0], []]
    bufs = [native.ints_to_limbs(p).reshape(-1, 4) if p else np.zeros((0, 4), dtype=np.uint64) for p in polys]
    assert kzg.commit(ck, bufs) == kzg.commit(ck, polys)
    assert kzg.commit(ck, [bufs[0], polys[1]]) == kzg.commit(ck, polys[:2])                      # mixed forms in one call
    z, xi = rng.randrange(cv.r), rng.randrange(cv.r)
    assert kzg.open(ck, bufs[:3], z, xi) == kzg.open(ck, polys[:3], z, xi)
    too_long = native.ints_to_limbs([1] * 65)
    with pytest.raises(ValueError):
        kzg.commit(ck, [too_long])
    padded = np.concatenate([bufs[0], np.zeros((9, 4), dtype=np.uint64)])                         # degree 63 with zero padding
    assert kzg.commit(ck, [padded]) == kzg.commit(ck, [polys[0]])


@pytest.mark.parametrize("curve,n", [("bn254", (1 << 18) + 5), ("bls12_381", 1 << 20)])
def test_caller_supplied_key_at_size_through_srs_load_g1(native, kzgs, curve, n):
    """kzg_srs_load_g1 is the entry a reference-side binding calls with its `ck` list (kzg.py:80,112-116).  At
    the sizes that take the 20-bit-window table: export a generated key, set a few points to infinity (flag) and
    make a few duplicates, import the arrays, commit, and compare with the trapdoor value of the EDITED key:
    sum_i c_i * tau^e(i) * G1 with the removed terms left out and duplicated points counted at their source's
    exponent.  Also re-exports the imported key and compares it with what went in."""
    cv = O.curve(curve)
    r = cv.r
    ctx = native.get_context(curve)
    tau = 0x5eed5eed5eed5eed1234 % r
    gen = ctx.srs_generate(native.int_to_words(tau), n)
    xy, inf = gen.export()
    gen.close()
    assert inf.sum() == 0
    removed = [0, 7, 4097, n // 2, n - 1]                      # points at infinity in the caller's list
    dup = {5: 3, 100001: 100000, n - 2: 1}                     # ck[dst] = ck[src]
    for i in removed:
        inf[i] = 1
        xy[i] = 0
    for dst, src in dup.items():
        xy[dst] = xy[src]
    srs = ctx.srs_load_g1(xy, inf)
    xy2, inf2 = srs.export()
    assert np.array_equal(inf2, inf) and np.array_equal(xy2[inf == 0], xy[inf == 0])
    rs = np.random.RandomState(n % 997)
    raw = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64)
    raw[:, 3] >>= np.uint64(4)
    coeffs = native.limbs_to_ints(raw)
    got_xy, got_inf = ctx.commit(srs, raw.reshape(1, n, 4), [n], n)
    L = ctx.fp_limbs
    got = native.limbs_to_ints(got_xy.reshape(2, L))
    # exponent-wise: sum c_i tau^i, minus removed terms, with duplicates re-pointed
    total = O.poly_eval(coeffs, tau, r)
    for i in removed:
        total -= coeffs[i] * pow(tau, i, r)
    for dst, src in dup.items():
        total += coeffs[dst] * (pow(tau, src, r) - pow(tau, dst, r))
    want = O.normalize(O.multiply(O.from_affine(cv.g1), total % r, cv), cv)
    assert got_inf[0] == 0 and (got[0], got[1]) == want
    srs.close()


def test_commit_2_20_with_extreme_coefficients(native, kzgs):
    """Scalars that stress the 13-window signed-digit recode at the BASELINE size: r-1, the whole run
    r-2^20 .. r-1, values whose every 20-bit window is 2^19 (the digit that carries) or 2^20-1, 2^255-adjacent
    values reduced below r, and powers of two on window boundaries -- beside uniform ones over [0, r).  Trapdoor
    identity commit(ck, p) == p(tau)*G1 (kzg.py:112-116)."""
    cv = O.BLS12_381
    r = cv.r
    kzg = kzgs["bls12_381"]
    n = 1 << 20
    tau = 0x6b7a675f736e61726b7a675f736e6172 % r
    ck, _ = kzg.setup(n - 1, tau=tau)
    rs = np.random.RandomState(31)
    raw = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2)
    raw[:, 3] %= np.uint64(r >> 192)                                          # uniform-ish over [0, r)
    special = [r - 1 - i for i in range(1 << 12)]                             # r-1 downwards
    special += [r - (1 << 20) + i for i in range(1 << 12)]                    # r-2^20 upwards
    half = sum(1 << (20 * j + 19) for j in range(13)) % r                     # every window = 2^19
    ones = sum(((1 << 20) - 1) << (20 * j) for j in range(13)) % r            # every window = 2^20-1
    special += [half, ones, (1 << 255) % r, ((1 << 255) - 1) % r, ((1 << 255) - 19) % r, (1 << 254), (1 << 254) - 1]
    special += [1 << (20 * j) for j in range(13)] + [(1 << (20 * j)) - 1 for j in range(1, 13)]
    special += [(1 << (20 * j + 19)) for j in range(12)]
    pos = rs.choice(n, size=len(special), replace=False)
    raw[pos] = native.ints_to_limbs(special)
    coeffs = native.limbs_to_ints(raw)
    ctx = native.get_context("bls12_381")
    xy, inf = ctx.commit(ck.srs, raw.reshape(1, n, 4), [n], n)
    got = native.limbs_to_ints(xy.reshape(2, 6))
    assert inf[0] == 0 and (got[0], got[1]) == O.normalize(O.commit_trapdoor(coeffs, tau, cv), cv)
    # the special values alone (everything else zero: the zero-skip of kzg.py:113-114 at size)
    sparse = np.zeros_like(raw)
    sparse[pos] = raw[pos]
    xy, inf = ctx.commit(ck.srs, sparse.reshape(1, n, 4), [n], n)
    got = native.limbs_to_ints(xy.reshape(2, 6))
    sp = native.limbs_to_ints(sparse)
    assert inf[0] == 0 and (got[0], got[1]) == O.normalize(O.commit_trapdoor(sp, tau, cv), cv)


@pytest.mark.parametrize("curve", CURVES)
def test_open_at_the_tile_boundaries_of_its_kernels(native, kzgs, curve):
    """The combination groups six products per reduction (Field::dot) and takes at most sixteen polynomials per
    launch; the scan works on chunks of 8 coefficients and tiles of 1024 or 2048 (128 / 256 threads), every tile
    summing the aggregates of the tiles above it: k = 1 .. 18 polynomials (one to three groups, every remainder, and
    the pre-combined form beyond sixteen) of lengths around those sizes with both tile widths, coefficients r-1 / 0 /
    random, against the trapdoor identity open == ((P(tau) - P(z)) / (tau - z)) G1 and P(z) against the oracle's
    Horner (kzg.py:148-154)."""
    cv = O.curve(curve)
    r = cv.r
    kzg = kzgs[curve]
    tau = 0xfeedface12345 % r
    ck, _ = kzg.setup(4200, tau=tau)
    ctx = native.get_context(curve)
    rng = random.Random(1234)
    L = ctx.fp_limbs
    cases = ((1, 1), (2, 2), (7, 3), (255, 1), (256, 2), (257, 3), (1023, 4), (1024, 5), (1025, 6), (2049, 7), (9, 8),
             (8, 9), (513, 11), (2047, 13), (2048, 16), (4097, 17), (3071, 18), (4100, 2))
    try:
        for tb in (128, 256):
            ctx.set_tuning("open_tile_threads", tb)
            for n, k in cases:
                lens = [max(1, n - 3 * i) for i in range(k)]
                polys = []
                for i, m in enumerate(lens):
                    kind = (i + n) % 3
                    polys.append([r - 1] * m if kind == 0 else [rng.randrange(r) for _ in range(m)] if kind == 1
                                 else [(r - 1) if j % 2 else 0 for j in range(m)])
                arr = np.zeros((k, n, 4), dtype=np.uint64)
                for i, p in enumerate(polys):
                    arr[i, :len(p)] = native.ints_to_limbs(p)
                z, xi = (r - 1, r - 1) if n % 2 else (rng.randrange(r), rng.randrange(r))
                xy, inf, ev = ctx.open(ck.srs, arr, lens, n, native.int_to_words(z), native.int_to_words(xi))
                comb = O.combine(polys, xi, r)
                assert native.limbs_to_ints(ev.reshape(1, 4))[0] == O.poly_eval(comb, z, r), (tb, n, k)
                want = O.normalize(O.open_trapdoor(polys, z, xi, tau, cv), cv)
                got = None if inf[0] else tuple(native.limbs_to_ints(xy.reshape(2, L)))
                assert got == want, (tb, n, k)
    finally:
        ctx.set_tuning("open_tile_threads", 0)


@pytest.mark.parametrize("curve", CURVES)
def test_open_with_grouped_tile_aggregates(native, kzgs, curve):
    """Beyond 1024 tiles (polynomials of more than 2^21 coefficients) the aggregates of 64 tiles are folded into one
    before the tiles sum what lies above them.  The same path forced at a size the oracle walks in seconds
    (`open_direct_tiles` = 1): 150,001 and 65,536 coefficients = 147 / 64 tiles of 1024 in 3 / 1 groups and 74 / 32
    tiles of 2048; the quotient's commitment against the trapdoor identity, P(z) against the oracle, and both equal
    to what the ungrouped path returns."""
    cv = O.curve(curve)
    r = cv.r
    kzg = kzgs[curve]
    tau = 0x5eed5eed77 % r
    ck, _ = kzg.setup(150001, tau=tau)
    ctx = native.get_context(curve)
    rng = random.Random(99)
    L = ctx.fp_limbs
    try:
        for n, k in ((150001, 2), (65536, 3)):
            lens = [n - 5 * i for i in range(k)]
            polys = [[rng.randrange(r) for _ in range(m)] for m in lens]
            arr = np.zeros((k, n, 4), dtype=np.uint64)
            for i, p in enumerate(polys):
                arr[i, :len(p)] = native.ints_to_limbs(p)
            z, xi = rng.randrange(r), rng.randrange(r)
            comb = O.combine(polys, xi, r)
            want_ev = O.poly_eval(comb, z, r)
            want = O.normalize(O.open_trapdoor(polys, z, xi, tau, cv), cv)
            for tb in (128, 256):
                for direct in (1, 0):
                    ctx.set_tuning("open_tile_threads", tb)
                    ctx.set_tuning("open_direct_tiles", direct)
                    xy, inf, ev = ctx.open(ck.srs, arr, lens, n, native.int_to_words(z), native.int_to_words(xi))
                    assert native.limbs_to_ints(ev.reshape(1, 4))[0] == want_ev, (n, tb, direct)
                    got = None if inf[0] else tuple(native.limbs_to_ints(xy.reshape(2, L)))
                    assert got == want, (n, tb, direct)
    finally:
        ctx.set_tuning("open_tile_threads", 0)
        ctx.set_tuning("open_direct_tiles", 0)


@pytest.mark.parametrize("curve", CURVES)
def test_one_open_poly_span_per_opening(native, kzgs, curve):
    """kzg_prof_read("open_poly") counts ONE span per opening (round 3 opened two scopes and halved the stage time the
    bench reported), on the synchronous and on the pipelined entry point."""
    import torch
    kzg = kzgs[curve]
    r = O.curve(curve).r
    ck, _ = kzg.setup(3000, tau=12345)
    ctx = native.get_context(curve)
    rng = random.Random(5)
    n, k = 2500, 3
    arr = native.ints_to_limbs([rng.randrange(r) for _ in range(n * k)]).reshape(k, n, 4)
    d = torch.from_numpy(arr.view(np.int64)).cuda()
    z, xi = native.int_to_words(rng.randrange(r)), native.int_to_words(rng.randrange(r))
    ctx.prof_enable(True)
    try:
        ctx.prof_reset()
        for _ in range(3):
            ctx.open(ck.srs, d.data_ptr(), [n] * k, n, z, xi, device=True)
        outs = [(np.zeros(2 * ctx.fp_limbs, np.uint64), np.zeros(1, np.uint8), np.zeros(4, np.uint64)) for _ in range(4)]
        for o in outs:
            ctx.open_device_async(ck.srs, d.data_ptr(), [n] * k, n, z, xi, *o)
        ctx.commit_flush()
        ms, cnt = ctx.prof_read("open_poly")
        assert cnt == 7 and ms > 0
    finally:
        ctx.prof_enable(False)


@pytest.mark.parametrize("key_log", [15, 18])
def test_bins_at_the_edge_of_the_sort_stage(native, kzgs, key_log):
    """Partition 2 sorts a bin of <= 7168 entries in one workgroup (registers + LDS stage) and hands a larger one to
    the chunked kernels (csrc/msm_prep.hip): bins holding exactly 7168, 7169, 7167, 1 and 0 entries side by side, for
    the 16-bit-window path (keys below 2^18 points) and the 20-bit one.  A scalar v in [1, 2^15] is ONE digit, in
    window 0, bucket v-1, bin (v-1) >> 8; trapdoor identity on the result (kzg.py:112-116)."""
    cv = O.BLS12_381
    r = cv.r
    ctx = native.get_context("bls12_381")
    n_key = 1 << key_log
    tau = 0x1357924680 % r
    srs = ctx.srs_generate(native.int_to_words(tau), n_key)
    rs = np.random.RandomState(key_log)
    vals = []
    for b, cnt in ((0, 7168), (1, 7169), (2, 7167), (3, 1), (5, 300)):
        vals += list(rs.randint(256 * b + 1, 256 * b + 257, size=cnt))
    vals = np.array(vals, dtype=np.uint64)
    rs.shuffle(vals)
    n = len(vals) + 500                                                  # the tail: zero scalars (skipped, kzg.py:113-114)
    raw = np.zeros((n, 4), dtype=np.uint64)
    raw[:len(vals), 0] = vals
    xy, inf = ctx.commit(srs, raw.reshape(1, n, 4), [n], n)
    coeffs = [int(v) for v in raw[:, 0]]
    got = tuple(native.limbs_to_ints(xy.reshape(2, 6)))
    assert inf[0] == 0 and got == O.normalize(O.commit_trapdoor(coeffs, tau, cv), cv)
    # and negative digits of the same magnitudes (r - v): every window is non-zero, window 0 mirrors the above
    neg = native.ints_to_limbs([(r - int(v)) % r for v in raw[:, 0]])
    xy, inf = ctx.commit(srs, neg.reshape(1, n, 4), [n], n)
    got = tuple(native.limbs_to_ints(xy.reshape(2, 6)))
    want = O.normalize(O.commit_trapdoor([(r - c) % r for c in coeffs], tau, cv), cv)
    assert inf[0] == 0 and got == want
    srs.close()


@pytest.mark.parametrize("log_n", [21, 22])
def test_commit_at_the_bin_counts_of_larger_polynomials(native, kzgs, log_n):
    """Partition 1 takes 4096 / 8192 bins for up to 2^21 / 2^22 scalars (msm_prep.hip, lob_for), so that a bin still fits
    the sort stage: trapdoor identity for uniform scalars over the whole range and for a skewed mix, at 2^21 + 3
    (what a rank of the 8-way config-4 job commits) and 2^22."""
    cv = O.BLS12_381
    r = cv.r
    ctx = native.get_context("bls12_381")
    n = (1 << log_n) + (3 if log_n == 21 else 0)
    tau = 0x2468ace13579 % r
    srs = ctx.srs_generate(native.int_to_words(tau), n)
    rs = np.random.RandomState(log_n)
    raw = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2)
    raw += rs.randint(0, 2, size=(n, 4)).astype(np.uint64)
    raw[:, 3] %= np.uint64(r >> 192)
    for variant in ("uniform", "skewed"):
        if variant == "skewed":
            raw[: n // 4] = 0
            raw[n // 4: n // 2, 1:] = 0                                   # one-limb scalars: the low windows only
            raw[n // 2: n // 2 + 50000] = raw[n // 2]                     # one value 50,000 times: heavy buckets
        coeffs = native.limbs_to_ints(raw)
        xy, inf = ctx.commit(srs, raw.reshape(1, n, 4), [n], n)
        got = tuple(native.limbs_to_ints(xy.reshape(2, 6)))
        assert inf[0] == 0 and got == O.normalize(O.commit_trapdoor(coeffs, tau, cv), cv), (log_n, variant)
    srs.close()
