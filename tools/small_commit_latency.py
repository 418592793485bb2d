#!/usr/bin/env python3
"""Commit latency at small sizes (the reference's own fixture scale): one synchronous commit, and commits
issued back to back through the pipelined entry point.  Launch-bound below 2^14 (about 25 kernel launches per
commit and a host-side Horner + inversion), not throughput-bound."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kzg_snark_amd import _native
ctx = _native.Context("bls12_381")
rs = np.random.RandomState(1)
for logn in (4, 8, 10, 12, 14, 16, 18):
    n = 1 << logn
    raw = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64); raw[:, 3] >>= np.uint64(3)
    srs = ctx.srs_generate(_native.int_to_words(12345), n)
    d = torch.from_numpy(raw.view(np.int64)).to("cuda:0")
    ctx.commit_device(srs, d.data_ptr(), [n], n)
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.commit_device(srs, d.data_ptr(), [n], n)
    t1 = time.perf_counter()
    xs = [np.zeros((1, 2 * ctx.fp_limbs), dtype=np.uint64) for _ in range(20)]
    infs = [np.zeros(1, dtype=np.uint8) for _ in range(20)]
    t2 = time.perf_counter()
    for i in range(20):
        ctx.commit_device_async(srs, d.data_ptr(), [n], n, xs[i], infs[i])
    ctx.commit_flush()
    t3 = time.perf_counter()
    print("2^%d: sync %.3f ms/commit, pipelined %.3f ms/commit" % (logn, (t1 - t0) / 20 * 1e3, (t3 - t2) / 20 * 1e3))
