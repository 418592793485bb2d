#!/usr/bin/env python3
"""Creates and destroys contexts and keys around a commit and an open and prints the free device memory:
the library must give back everything it allocated (run on the GPU box)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kzg_snark_amd import _native
r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
n = 1 << 18
rs = np.random.RandomState(1)
raw = rs.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64); raw[:, 3] >>= np.uint64(3)
free0 = None
for it in range(12):
    ctx = _native.Context("bls12_381")
    srs = ctx.srs_generate(_native.int_to_words(12345), n)
    xy, inf = ctx.commit(srs, raw.reshape(1, n, 4), [n], n)
    o = ctx.open(srs, raw.reshape(1, n, 4), [n], n, _native.int_to_words(5), _native.int_to_words(7))
    srs.close(); ctx.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 1: free0 = free
    print(it, free >> 20, "MiB free")
print("leak per iteration (MiB):", (free0 - free) / 10 / 2**20)
