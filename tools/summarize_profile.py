#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh output directory (gpurun_out/prof_<tag>) into
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.csv and profiles/traffic.json.

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB, collected in
separate --pmc passes; on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane loads, so
read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections
import csv
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
    m = re.search(r"(\w+_kernel)<?", name)
    if "rocprim" in name:
        return "rocprim_scan"
    return m.group(1) if m else name[:48]


rows = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,total_ns,avg_ns,percent,min_ns,max_ns\n")
    for r in rows:
        f.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},"
                f"{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
avg_ns = collections.defaultdict(list)
for r in rows:
    avg_ns[short(r["Name"])].append((int(r["Calls"]), float(r["AverageNs"])))

pmc = {}
for ctr, sub, fn in (("FETCH_SIZE", "pmc_fetch", "fetch_counter_collection.csv"),
                     ("WRITE_SIZE", "pmc_write", "write_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(os.path.join(src, sub, fn))):
        if row["Counter_Name"] == ctr:
            agg[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    pmc[ctr] = {k: sum(v) / len(v) for k, v in agg.items()}
with open(os.path.join(dst, f"{tag}_pmc.csv"), "w") as f:
    f.write("kernel,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,hbm_read_bytes(2x FETCH),hbm_write_bytes,hbm_bytes_per_launch\n")
    for k in sorted(set(pmc["FETCH_SIZE"]) | set(pmc["WRITE_SIZE"])):
        fk, wk = pmc["FETCH_SIZE"].get(k, 0.0), pmc["WRITE_SIZE"].get(k, 0.0)
        f.write(f"{k},{fk:.1f},{wk:.1f},{2 * fk * 1024:.0f},{wk * 1024:.0f},{(2 * fk + wk) * 1024:.0f}\n")


def traffic(k):
    return (2 * pmc["FETCH_SIZE"].get(k, 0.0) + pmc["WRITE_SIZE"].get(k, 0.0)) * 1024


out = {
    "source": f"profiles/{tag}_pmc.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; "
              "read bytes = 2 x FETCH_SIZE KiB x 1024 per MI355X_MICROARCH.md, gfx950 correction)",
    "msm_accumulate_kernel": traffic("msm_accumulate_kernel"),
    "ntt_pass_kernel_per_launch": traffic("ntt_pass_kernel"),
    "ntt_batch_per_launch": batch,
    "ntt_pass_kernel_per_transform": 2 * traffic("ntt_pass_kernel") / batch,
}
json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "trace_bench.log"), os.path.join(dst, f"{tag}_bench_under_rocprof.log"))
print(json.dumps(out, indent=1))
