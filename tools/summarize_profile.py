#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh output directory (gpurun_out/prof_<tag>) into
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.csv, profiles/<tag>_ntt_alone_kernel_stats.csv and
profiles/counters.json (what bench.py reports as roofline.traffic / roofline.limiter).

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB, collected in
separate --pmc passes; on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane loads, so
read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-B-per-lane stores.
SQ_INSTS_VALU counts wave-instructions (one per wave64 instruction issued), summed over the launch."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4     # transforms per ntt_pass launch in bench.py
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    if "rocprim" in name:
        return "rocprim_scan"
    m = re.search(r"(\w+_kernel)<?", name)
    return m.group(1) if m else name[:48]


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


def stats(path, out_name):
    rows = list(csv.DictReader(open(path)))
    with open(os.path.join(dst, out_name), "w") as f:
        f.write("kernel,calls,total_ns,avg_ns,percent,min_ns,max_ns\n")
        for r in rows:
            f.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},"
                    f"{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
    out = collections.defaultdict(list)      # a kernel template may appear as several rows (one per instantiation)
    for r in rows:
        out[short(r["Name"])].append((float(r["AverageNs"]), int(r["Calls"])))
    return out


avg_ns = stats(one("trace/**/*kernel_stats.csv"), f"{tag}_kernel_stats.csv")
ntt_stats = one("ntt_alone/**/*kernel_stats.csv")
ntt_alone_ns = stats(ntt_stats, f"{tag}_ntt_alone_kernel_stats.csv") if ntt_stats else {}

pmc = collections.defaultdict(dict)          # counter -> kernel -> average per launch
for sub in ("pmc_fetch", "pmc_write", "pmc_valu", "pmc_mem"):
    path = one(sub + "/**/*counter_collection.csv")
    if not path:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(path)):
        agg[row["Counter_Name"]][short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    for ctr, per_kernel in agg.items():
        for k, v in per_kernel.items():
            pmc[ctr][k] = sum(v) / len(v)
counters = sorted(pmc)
kernels = sorted({k for c in pmc.values() for k in c})
with open(os.path.join(dst, f"{tag}_pmc.csv"), "w") as f:
    f.write("kernel," + ",".join(f"{c}_avg_per_launch" for c in counters) + ",hbm_read_bytes(2x FETCH KiB),hbm_write_bytes,hbm_bytes_per_launch\n")
    for k in kernels:
        fk, wk = pmc.get("FETCH_SIZE", {}).get(k, 0.0), pmc.get("WRITE_SIZE", {}).get(k, 0.0)
        f.write(k + "," + ",".join(f"{pmc[c].get(k, 0.0):.6g}" for c in counters)
                + f",{2 * fk * 1024:.0f},{wk * 1024:.0f},{(2 * fk + wk) * 1024:.0f}\n")


def traffic(k):
    return (2 * pmc.get("FETCH_SIZE", {}).get(k, 0.0) + pmc.get("WRITE_SIZE", {}).get(k, 0.0)) * 1024


def valu(k):
    return pmc.get("SQ_INSTS_VALU", {}).get(k)


out = {
    "source": f"profiles/{tag}_pmc.csv (rocprofv3 --kernel-trace --pmc, one counter group per run: FETCH_SIZE | WRITE_SIZE | "
              "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES; read bytes = 2 x FETCH_SIZE KiB x 1024, the gfx950 correction of "
              "MI355X_MICROARCH.md; bench.py --mode batch --steps 3)",
    "msm_accumulate_kernel": {"hbm_bytes": traffic("msm_accumulate_kernel"),
                              "valu_wave_instructions": valu("msm_accumulate_kernel"),
                              "avg_ns_pipelined_under_profiler": (avg_ns.get("msm_accumulate_kernel") or [(None, 0)])[0][0]},
    "ntt_pass_kernel_per_transform": {"hbm_bytes": 2 * traffic("ntt_pass_kernel") / batch,
                                      "valu_wave_instructions": (2 * valu("ntt_pass_kernel") / batch) if valu("ntt_pass_kernel") else None,
                                      "ntt_batch_per_launch": batch,
                                      # the two passes of a transform are two instantiations (twist / reduce epilogue):
                                      # one launch of each per batch of transforms
                                      "avg_ns_per_transform_alone_under_profiler":
                                          (sum(a for a, _ in ntt_alone_ns.get("ntt_pass_kernel", [])) / batch) or None,
                                      "avg_ns_per_launch_alone_under_profiler":
                                          [a for a, _ in ntt_alone_ns.get("ntt_pass_kernel", [])]},
}
json.dump(out, open(os.path.join(dst, "counters.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "trace_bench.log"), os.path.join(dst, f"{tag}_bench_under_rocprof.log"))
print(json.dumps(out, indent=1))
