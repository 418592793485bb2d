#!/usr/bin/env python3
"""Commit timing for skewed scalar distributions at 2^20 (uniform vs all-ones vs {0,1,2} vs 16-bit)."""
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
cases = {}
u = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64); u[:, 3] >>= np.uint64(3)
cases["uniform"] = u
a = np.zeros((n, 4), dtype=np.uint64); a[:, 0] = 1
cases["all_ones"] = a
s = np.zeros((n, 4), dtype=np.uint64); s[:, 0] = rs.randint(0, 3, size=n)
cases["zero_one_two"] = s
h = np.zeros((n, 4), dtype=np.uint64); h[:, 0] = rs.randint(0, 1 << 16, size=n)
cases["16_bit"] = h
for name, arr in cases.items():
    d = torch.from_numpy(arr.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    ctx.commit_device(srs, d.data_ptr(), [n], n)
    t0 = time.perf_counter()
    for _ in range(3):
        xy, inf = ctx.commit_device(srs, d.data_ptr(), [n], n)
    print(name, "%.2f ms per commit" % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
