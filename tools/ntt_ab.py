#!/usr/bin/env python3
"""Same-box A/B timing of the NTT across library builds, alternating, one fresh process per measurement.

    python tools/ntt_ab.py [--log-n 20] [--batch 4] [--rounds 3] name=path/to/libkzg_mi355x.so[,ENV=VAL...] ...

Each child times `iters` batched transforms of 2^log_n (forward / inverse alternating, in place, data resident in
HBM) by wall clock around a synchronised loop and by the library's own HIP events (span "ntt_pass"), and checks the
result of one forward + inverse round trip.  Prints microseconds per transform."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(log_n, batch, iters):
    import time
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    sys.path.insert(0, ROOT)
    import torch
    from kzg_snark_amd import _native
    r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    w = pow(7, (r - 1) >> log_n, r)
    ctx = _native.Context("bls12_381")
    g = torch.Generator().manual_seed(7)
    x = torch.randint(0, 1 << 62, (batch, 1 << log_n, 4), dtype=torch.int64, generator=g)
    x[:, :, 3] >>= 3
    d = x.to("cuda:0")
    ref = d.clone()
    torch.cuda.synchronize()
    ww = _native.int_to_words(w)
    for i in range(4):
        ctx.ntt_device(d.data_ptr(), log_n, ww, bool(i & 1), batch)
    ctx.synchronize()
    ok = bool(torch.equal(d, ref))
    ctx.prof_enable(True)
    ctx.prof_reset()
    t0 = time.perf_counter()
    for i in range(iters):
        ctx.ntt_device(d.data_ptr(), log_n, ww, bool(i & 1), batch)
    ctx.synchronize()
    wall = time.perf_counter() - t0
    ms, cnt = ctx.prof_read("ntt_pass")
    print(json.dumps({"wall_us": wall / (iters * batch) * 1e6, "event_us": ms * 1e3 / (iters * batch), "launches": cnt,
                      "round_trip_ok": ok}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    if a.child:
        return child(a.log_n, a.batch, a.iters)
    rows = {}
    for rnd in range(a.rounds):
        for v in a.variants:
            name, rest = v.split("=", 1)
            parts = rest.split(",")
            env = dict(os.environ, KZG_MI355X_LIB=os.path.abspath(parts[0]))
            for kv in parts[1:]:
                k, val = kv.split("=", 1)
                env[k] = val
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--log-n", str(a.log_n), "--batch",
                                  str(a.batch), "--iters", str(a.iters)], env=env, capture_output=True, text=True)
            try:
                rows.setdefault(name, []).append(json.loads(out.stdout.strip().splitlines()[-1]))
            except Exception:
                rows.setdefault(name, []).append({"error": (out.stderr or out.stdout)[-300:]})
            print(f"round {rnd} {name}: {rows[name][-1]}", flush=True)
    print(f"\n2^{a.log_n} NTT, batch {a.batch}, {a.iters} launches per measurement, us per transform (HIP events | wall), "
          f"rounds alternate on one box")
    for name, rs in rows.items():
        ev = " ".join(f"{r['event_us']:7.2f}" if "event_us" in r else "  error" for r in rs)
        wl = " ".join(f"{r['wall_us']:7.2f}" if "wall_us" in r else "  error" for r in rs)
        ok = all(r.get("round_trip_ok") for r in rs)
        print(f"{name:28s} events {ev}   wall {wl}   round trip {'ok' if ok else 'FAILED'}")


if __name__ == "__main__":
    main()
