#!/usr/bin/env python3
"""Opening-only driver for profiling: `iters` KZG.open calls of k polynomials of 2^log_n coefficients
(kzg.py:122-159; plonk/prover.py:184 opens k = 6), device-resident inputs, synchronous entry point -- so every
tile_combine_kernel / tile_fill_kernel launch a profiler sees belongs to an opening.

    python tools/open_only.py [log_n=20] [k=6] [iters=10] [async]

`async`: the same openings through kzg_open_device_async + kzg_commit_flush (witness MSMs pipelined, the polynomial
stage of opening p + 1 beside the accumulate kernel of opening p); prints opens/s."""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kzg_snark_amd import _native  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n = 1 << log_n
ctx = _native.Context("bls12_381")
srs = ctx.srs_generate(_native.int_to_words(0x6b7a675f736e6172), n)
g = torch.Generator().manual_seed(11)
x = torch.randint(0, 1 << 62, (k, n, 4), dtype=torch.int64, generator=g)
x[:, :, 3] >>= 3
d = x.to("cuda:0")
torch.cuda.synchronize()
lens = [n - i for i in range(k)]
z, xi = _native.int_to_words(0x1111111111111111111111111111), _native.int_to_words(0x2222222222222222222222)
if len(sys.argv) > 4 and sys.argv[4] == "async":
    import time
    import numpy as np
    L = ctx.fp_limbs
    outs = [(np.zeros(2 * L, np.uint64), np.zeros(1, np.uint8), np.zeros(4, np.uint64)) for _ in range(iters)]
    for o in outs[:4]:
        ctx.open_device_async(srs, d.data_ptr(), lens, n, z, xi, *o)
    ctx.commit_flush()
    ctx.prof_enable(True)
    ctx.prof_reset()
    t0 = time.perf_counter()
    for o in outs:
        ctx.open_device_async(srs, d.data_ptr(), lens, n, z, xi, *o)
    ctx.commit_flush()
    dt = time.perf_counter() - t0
    ms, cnt = ctx.prof_read("open_poly")
    print(f"pipelined: {iters / dt:.1f} opens/s ({dt / iters * 1e3:.3f} ms per opening); open_poly span beside the MSMs "
          f"{ms / max(cnt, 1) * 1e3:.1f} us over {cnt} openings (k = {k}, 2^{log_n})")
    sys.exit(0)
ctx.prof_enable(True)
for i in range(iters):
    if i == 2:
        ctx.prof_reset()
    ctx.open(srs, d.data_ptr(), lens, n, z, xi, device=True)
ms, cnt = ctx.prof_read("open_poly")
print(f"open_poly: {ms / max(cnt, 1) * 1e3:.1f} us per opening over {cnt} openings (HIP events; k = {k}, 2^{log_n})")
