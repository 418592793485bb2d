#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (not part of the test suite: a few minutes of random shapes and scalar
distributions).  NTT / INTT / ragged fft_ff against the C restatement of fft_ff.py; commit and open against the
trapdoor identities of a key generated with a known tau.

    python tools/fuzz_gpu.py [seconds=120] [seed=1]"""
import os
import random
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from kzg_snark_amd import _native as N  # noqa: E402
from oracle import c_oracle as CO, py_oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
rs = np.random.RandomState(seed)
t_end = time.time() + budget
stats = {"ntt": 0, "ragged": 0, "commit": 0, "open": 0, "open_shard": 0}
keys = {}


def scalars(n, r, kind):
    if kind == "uniform":
        raw = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2) + rs.randint(0, 2, size=(n, 4)).astype(np.uint64)
        raw[:, 3] %= np.uint64(r >> 192)
        return raw
    if kind == "small":
        raw = np.zeros((n, 4), dtype=np.uint64)
        raw[:, 0] = rs.randint(0, rng.choice([2, 3, 256, 1 << 16, 1 << 20, 1 << 21]), size=n)
        return raw
    if kind == "sparse":
        raw = scalars(n, r, "uniform")
        raw[rs.rand(n) < 0.9] = 0
        return raw
    if kind == "repeated":
        raw = scalars(n, r, "uniform")
        raw[:] = raw[rs.randint(0, max(1, min(n, 5)), size=n)]
        return raw
    if kind == "near_r":
        return N.ints_to_limbs([(r - 1 - rng.randrange(1 << rng.choice([1, 8, 20, 40]))) % r for _ in range(n)])
    raise ValueError(kind)


while time.time() < t_end:
    curve = rng.choice(["bls12_381", "bn254"])
    cv = O.curve(curve)
    r = cv.r
    ctx = N.get_context(curve)
    what = rng.choice(["ntt", "ntt", "ragged", "commit", "commit", "open", "open", "open_shard"])
    if what == "ntt":
        log_n = rng.randrange(1, 17)
        n = 1 << log_n
        w = rng.choice([cv.root_of_unity(n), rng.randrange(r), r - 1, 1])
        inverse = rng.random() < 0.5
        if inverse and w == 0:
            continue
        raw = scalars(n, r, rng.choice(["uniform", "small", "near_r"]))
        got = raw.copy()
        ctx.ntt(got, log_n, N.int_to_words(w), inverse)
        want = raw.copy()
        CO.fft(curve, want, w, inverse=inverse)
        assert np.array_equal(got, want), ("ntt", curve, log_n, hex(w), inverse)
    elif what == "ragged":
        n = rng.randrange(2, 5000)
        w = rng.randrange(1, r)
        inverse = rng.random() < 0.5
        raw = scalars(n, r, rng.choice(["uniform", "near_r"]))
        got = raw.copy()
        ctx.fft_ff_any(got, N.int_to_words(w), inverse)
        want = raw.copy()
        CO.fft(curve, want, w, inverse=inverse)
        assert np.array_equal(got, want), ("ragged", curve, n, inverse)
    else:
        key_n = rng.choice([64, 1000, 5000, 40000, 1 << 18])
        if (curve, key_n) not in keys:
            if len(keys) >= 4:
                keys.pop(next(iter(keys)))[0].close()
            tau = rng.randrange(2, r)
            keys[(curve, key_n)] = (ctx.srs_generate(N.int_to_words(tau), key_n), tau)
        srs, tau = keys[(curve, key_n)]
        L = ctx.fp_limbs
        g = O.from_affine(cv.g1)
        if what == "commit":
            k = rng.randrange(1, 6)
            stride = rng.randrange(1, key_n + 1)
            lens = [rng.randrange(0, stride + 1) for _ in range(k)]
            arr = np.zeros((k, stride, 4), dtype=np.uint64)
            for i, m in enumerate(lens):
                if m:
                    arr[i, :m] = scalars(m, r, rng.choice(["uniform", "small", "sparse", "repeated", "near_r"]))
            xy, inf = ctx.commit(srs, arr, lens, stride)
            for i, m in enumerate(lens):
                coeffs = N.limbs_to_ints(arr[i, :m]) if m else []
                want = O.normalize(O.multiply(g, O.poly_eval(coeffs, tau, r), cv), cv) if coeffs else None
                got = None if inf[i] else tuple(N.limbs_to_ints(xy[i].reshape(2, L)))
                assert got == want, ("commit", curve, key_n, stride, lens, i)
        elif what == "open_shard":
            # one polynomial set cut into G contiguous coefficient ranges, each opened as a shard (slice evaluation,
            # carry, quotient slice against key points lo-1 ..): the partial proofs add up to the trapdoor value
            import torch
            k = rng.randrange(1, 8)
            n = rng.randrange(8, min(key_n, 30000) + 1)
            G = rng.randrange(1, 5)
            cuts = sorted(rng.sample(range(1, n), min(G - 1, n - 1)))
            bounds = [0] + cuts + [n]
            arr = np.zeros((k, n, 4), dtype=np.uint64)
            polys = []
            for i in range(k):
                arr[i] = scalars(n, r, rng.choice(["uniform", "small", "near_r"]))
                polys.append(N.limbs_to_ints(arr[i]))
            z, xi = rng.choice([0, 1, r - 1, rng.randrange(r)]), rng.choice([1, rng.randrange(r)])
            if z == tau:
                continue
            ctx.set_tuning("open_tile_threads", rng.choice([0, 128, 256]))
            ctx.set_tuning("open_direct_tiles", rng.choice([0, 0, 1, 2]))
            zw, xw = N.int_to_words(z), N.int_to_words(xi)
            H, acc, ev0 = [], O.Z1(), None
            shards = []
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                t = torch.from_numpy(np.ascontiguousarray(arr[:, lo:hi]).view(np.int64)).cuda()
                H.append(N.limbs_to_ints(ctx.open_shard_begin(t.data_ptr(), [hi - lo] * k, hi - lo, zw, xw).reshape(1, 4))[0])
                shards.append((lo, hi))
            for gi, (lo, hi) in enumerate(shards):
                carry = sum(H[h] * pow(z, shards[h][0] - hi, r) for h in range(gi + 1, len(shards))) % r
                t = torch.from_numpy(np.ascontiguousarray(arr[:, lo:hi]).view(np.int64)).cuda()
                ctx.open_shard_begin(t.data_ptr(), [hi - lo] * k, hi - lo, zw, xw)        # a context holds one slice
                start = 0 if gi == 0 else lo - 1
                cnt = hi - 1 - start                    # 0: a first slice of one coefficient commits nothing, but still
                ks = ctx.srs_generate(N.int_to_words(tau), max(cnt, 1), start=start)      # reports P(z)
                xy, inf, ev = ctx.open_shard_finish(ks, zw, N.int_to_words(carry), gi == 0)
                ks.close()
                if gi == 0:
                    ev0 = N.limbs_to_ints(ev.reshape(1, 4))[0]
                if not inf[0]:
                    acc = O.add(acc, O.from_affine(tuple(N.limbs_to_ints(xy.reshape(2, L)))), cv)
            ctx.set_tuning("open_tile_threads", 0)
            ctx.set_tuning("open_direct_tiles", 0)
            comb = O.combine(polys, xi, r)
            assert ev0 == O.poly_eval(comb, z, r), ("open_shard eval", curve, n, bounds)
            assert O.normalize(acc, cv) == O.normalize(O.open_trapdoor(polys, z, xi, tau, cv), cv), ("open_shard", curve, n, bounds, z)
        else:
            k = rng.choice([rng.randrange(1, 9), rng.randrange(9, 21)])
            ctx.set_tuning("open_tile_threads", rng.choice([0, 128, 256]))
            ctx.set_tuning("open_direct_tiles", rng.choice([0, 0, 1, 3]))
            stride = rng.randrange(2, min(key_n, 20000) + 1)
            lens = [rng.randrange(1, stride + 1) for _ in range(k)]
            arr = np.zeros((k, stride, 4), dtype=np.uint64)
            polys = []
            for i, m in enumerate(lens):
                arr[i, :m] = scalars(m, r, rng.choice(["uniform", "small", "near_r"]))
                polys.append(N.limbs_to_ints(arr[i, :m]))
            z, xi = rng.choice([0, 1, r - 1, rng.randrange(r)]), rng.choice([0, 1, rng.randrange(r)])
            if z == tau:
                continue
            xy, inf, ev = ctx.open(srs, arr, lens, stride, N.int_to_words(z), N.int_to_words(xi))
            comb = O.combine(polys, xi, r)
            assert N.limbs_to_ints(ev.reshape(1, 4))[0] == O.poly_eval(comb, z, r), ("open eval", curve, lens)
            want = O.normalize(O.open_trapdoor(polys, z, xi, tau, cv), cv)
            got = None if inf[0] else tuple(N.limbs_to_ints(xy.reshape(2, L)))
            assert got == want, ("open", curve, key_n, stride, lens, z, xi)
            ctx.set_tuning("open_tile_threads", 0)
            ctx.set_tuning("open_direct_tiles", 0)
    stats[what] += 1
print("fuzz ok:", stats, "seed", seed)
