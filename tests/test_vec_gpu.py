"""Device vector / polynomial primitives (kzg_fr_*) against Python integers.  Bit-exact."""
import random

import numpy as np
import pytest

from oracle import py_oracle as O

pytestmark = pytest.mark.gpu


def dev(native, vals):
    import torch
    return torch.from_numpy(native.ints_to_limbs(vals).view(np.int64)).to("cuda:0")


def host(native, t):
    return native.limbs_to_ints(t.cpu().numpy().view(np.uint64))


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 1000, 1025, 40000])
def test_vector_primitives(native, curve, n):
    import torch
    cv = O.curve(curve)
    r = cv.r
    ctx = native.get_context(curve)
    rng = random.Random(n)
    a = [rng.randrange(r) for _ in range(n)]
    b = [rng.randrange(r) for _ in range(n)]
    if n > 2:
        a[1], b[2] = 0, r - 1
    da, db = dev(native, a), dev(native, b)
    out = torch.empty_like(da)
    for op, f in (("add", lambda x, y: (x + y) % r), ("sub", lambda x, y: (x - y) % r), ("mul", lambda x, y: x * y % r)):
        ctx.vec_op(op, n, da.data_ptr(), db.data_ptr(), out.data_ptr())
        ctx.synchronize()
        assert host(native, out) == [f(x, y) for x, y in zip(a, b)], op
    s, c0 = rng.randrange(r), rng.randrange(r)
    ctx.vec_mul_powers(n, da.data_ptr(), s, c0, out.data_ptr())
    ctx.synchronize()
    assert host(native, out) == [x * c0 * pow(s, i, r) % r for i, x in enumerate(a)]
    ctx.vec_inverse(n, da.data_ptr(), out.data_ptr())
    ctx.synchronize()
    assert host(native, out) == [pow(x, -1, r) if x else 0 for x in a]
    inpl = dev(native, a)                                  # in place (out aliases the input); a blocking upload:
                                                           # a clone() would run on torch's stream, not the library's
    ctx.vec_inverse(n, inpl.data_ptr(), inpl.data_ptr())
    ctx.synchronize()
    assert host(native, inpl) == [pow(x, -1, r) if x else 0 for x in a]
    if n >= 33:                                            # a whole chunk of zeros, and zeros at chunk borders
        az = list(a)
        for i in list(range(0, 32)) + [32, n - 1]:
            az[i] = 0
        dz = dev(native, az)
        ctx.vec_inverse(n, dz.data_ptr(), out.data_ptr())
        ctx.synchronize()
        assert host(native, out) == [pow(x, -1, r) if x else 0 for x in az]
    ctx.vec_prefix_product(n, db.data_ptr(), out.data_ptr())
    ctx.synchronize()
    want, acc = [], 1
    for y in b:
        want.append(acc)
        acc = acc * y % r
    assert host(native, out) == want
    z = rng.randrange(r)
    assert ctx.poly_eval(n, da.data_ptr(), z) == O.poly_eval(a, z, r)
    assert ctx.poly_eval(n, da.data_ptr(), 0) == a[0]
    # lincomb of three vectors of different lengths
    c = [rng.randrange(r) for _ in range(max(1, n // 2))]
    dc = dev(native, c)
    sc = [rng.randrange(r), 1, r - 1]
    ctx.vec_lincomb(n, [da.data_ptr(), db.data_ptr(), dc.data_ptr()], [n, n, len(c)], sc, out.data_ptr())
    ctx.synchronize()
    assert host(native, out) == [(sc[0] * a[i] + sc[1] * b[i] + sc[2] * (c[i] if i < len(c) else 0)) % r for i in range(n)]


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_poly_eval_at_the_tile_boundaries(native, curve):
    """kzg_fr_poly_eval = the first tile pass of the opening's scan without its stores + the sum of the tile
    aggregates: lengths around the chunk (8) and tile (2048) sizes and far beyond, z = 0 / 1 / r-1 / random,
    coefficients r-1 / random, also with the aggregates of 64 tiles folded first (`open_direct_tiles` = 1: the path
    polynomials beyond 2^21 coefficients take)."""
    r = O.curve(curve).r
    ctx = native.get_context(curve)
    rng = random.Random(8)
    try:
        for n in (1, 2, 7, 8, 9, 2047, 2048, 2049, 4096, 100001, 300000):
            a = [r - 1] * n if n % 2 == 0 and n < 5000 else [rng.randrange(r) for _ in range(n)]
            da = dev(native, a)
            for z in (0, 1, r - 1, rng.randrange(r)):
                want = O.poly_eval(a, z, r)
                for direct in ((0, 1) if n > 4096 else (0,)):
                    ctx.set_tuning("open_direct_tiles", direct)
                    assert ctx.poly_eval(n, da.data_ptr(), z) == want, (n, z, direct)
    finally:
        ctx.set_tuning("open_direct_tiles", 0)
