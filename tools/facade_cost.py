#!/usr/bin/env python3
"""What a reference-style call costs end to end at 2^20: fft_ff(list of field elements, w, F) and KZG.commit(ck, [list])
through the facade, split into Python-object marshalling and everything else, next to the buffer fast path."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import random
import numpy as np
from kzg_snark_amd import _native
from kzg_snark_amd.kzg import KZG
from kzg_snark_amd.fft_ff import fft_ff

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
kzg = KZG("bls12_381")
F, r = kzg.Fq, kzg.curve_order
w = F.root_of_unity(n)
rng = random.Random(2)
ints = [rng.randrange(r) for _ in range(n)]
elems = [F(v) for v in ints]
limbs = _native.ints_to_limbs(ints)
ck, _ = kzg.setup(n - 1, tau=12345)


def timed(label, fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    print("%-58s %8.1f ms" % (label, (time.perf_counter() - t0) / reps * 1e3), flush=True)
    return out


timed("fft_ff(list of F elements)  -> list of F elements", lambda: fft_ff(elems, w, F))
timed("fft_ff(list of ints)        -> list of F elements", lambda: fft_ff(ints, w, F))
timed("fft_ff(uint64[n,4] buffer)  -> buffer", lambda: fft_ff(limbs, w, F))
timed("  of which: ints_to_limbs", lambda: _native.ints_to_limbs(ints))
timed("  of which: limbs_to_ints", lambda: _native.limbs_to_ints(limbs))
timed("  of which: [F(v) for v in ints]", lambda: [F(v) for v in ints])
timed("KZG.commit(ck, [list of ints])", lambda: kzg.commit(ck, [ints]))
timed("KZG.commit(ck, [uint64[n,4] buffer])", lambda: kzg.commit(ck, [limbs]))
