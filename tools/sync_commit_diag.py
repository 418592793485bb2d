#!/usr/bin/env python3
"""Wall time and stage spans of SYNCHRONOUS 2^20 commits (one in flight) for two scalar distributions."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kzg_snark_amd import _native

n = 1 << 20
ctx = _native.Context("bls12_381")
srs = ctx.srs_generate(_native.int_to_words(12345), n)
rs = np.random.RandomState(1)
u62 = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64); u62[:, 3] >>= np.uint64(3)
full = (rs.randint(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2)
        + rs.randint(0, 2, size=(n, 4)).astype(np.uint64))
full[:, 3] %= np.uint64(0x73eda753299d7d48)
for name, arr in (("limbs<2^62", u62), ("full range", full)):
    d = torch.from_numpy(arr.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    ctx.commit_device(srs, d.data_ptr(), [n], n)
    ctx.prof_enable(True); ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.commit_device(srs, d.data_ptr(), [n], n)
    wall = (time.perf_counter() - t0) / 5 * 1e3
    spans = {k: ctx.prof_read(k) for k in ("msm_partition1", "msm_partition2", "msm_order", "msm_accumulate", "msm_finalize", "msm_reduce")}
    ctx.prof_enable(False)
    print(f"{name:11s} wall {wall:6.2f} ms per commit | " + "  ".join(f"{k[4:]} {v[0] / max(v[1], 1):.3f}" for k, v in spans.items()), flush=True)
