#!/usr/bin/env python3
"""NTT-only driver for profiling: `iters` forward/inverse 2^log_n transforms on a batch."""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kzg_snark_amd import _native  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
w = pow(7, (r - 1) >> log_n, r)
ctx = _native.Context("bls12_381")
x = torch.randint(0, 1 << 62, (batch, 1 << log_n, 4), dtype=torch.int64)
x[:, :, 3] >>= 3
d = x.to("cuda:0")
torch.cuda.synchronize()
ww = _native.int_to_words(w)
for i in range(iters):
    ctx.ntt_device(d.data_ptr(), log_n, ww, bool(i & 1), batch)
ctx.synchronize()
print("done")
