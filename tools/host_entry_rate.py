#!/usr/bin/env python3
"""Commit throughput through the HOST entry point (kzg_commit: scalars in host memory, one H2D copy per call) next to
the device entry point -- the PCIe-inclusive figure DESIGN.md quotes beside the headline (which is device-resident)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kzg_snark_amd import _native
log_n, B, iters = 20, 4, 10
n = 1 << log_n
ctx = _native.Context("bls12_381")
srs = ctx.srs_generate(_native.int_to_words(0x6b7a675f736e6172), n)
rs = np.random.RandomState(3)
host = rs.randint(0, 1 << 62, size=(B, n, 4)).astype(np.uint64); host[:, :, 3] >>= np.uint64(3)
pinned = torch.from_numpy(host.view(np.int64)).pin_memory()
dev = torch.from_numpy(host.view(np.int64)).to("cuda:0")
lens = [n] * B
for name, fn in (("host entry, pageable numpy", lambda: ctx.commit(srs, host, lens, n)),
                 ("host entry, pinned memory", lambda: ctx.commit(srs, pinned.numpy().view(np.uint64), lens, n)),
                 ("device entry", lambda: ctx.commit_device(srs, dev.data_ptr(), lens, n))):
    fn()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    dt = (time.perf_counter() - t0) / (iters * B)
    print("%-28s %.3f ms per commit  (%.0f commits/s; batch of %d per call, synchronous calls)" % (name, dt * 1e3, 1 / dt, B))
