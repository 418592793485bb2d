#!/usr/bin/env python3
"""Condenses tools/profile_open.sh's output (gpurun_out/prof_open_<tag>) into profiles/<tag>_open_kernel_stats.csv and
profiles/<tag>_open_pmc.csv and adds the entry "open_poly" to profiles/counters.json (what bench.py reports as
open.roofline.traffic): HBM bytes and VALU wave-instructions PER OPENING, i.e. summed over the launches one opening
issues (round 4: tile_combine + tile_fill; rounds 1-3: 1 lincomb + the levels of chunk_eval / chunk_fill + top_suffix).  Units and the gfx950 FETCH_SIZE correction
as in tools/summarize_profile.py (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_open_" + tag)
dst = os.path.join(ROOT, "profiles")
POLY = ("lincomb_kernel", "lincomb_eval_kernel", "chunk_eval_kernel", "chunk_fill_kernel", "chunk_fill_final_kernel",
        "top_suffix_kernel", "tile_combine_kernel", "tile_group_kernel", "tile_fill_kernel", "tile_eval_kernel")
FIRST = ("lincomb_kernel", "lincomb_eval_kernel", "tile_combine_kernel")      # one launch of these per opening


def short(name):
    m = re.search(r"(\w+_kernel)<?", name)
    return m.group(1) if m else name[:48]


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


rows = list(csv.DictReader(open(one("trace/**/*kernel_stats.csv"))))
n_open = sum(int(r["Calls"]) for r in rows if short(r["Name"]) in FIRST)
with open(os.path.join(dst, f"{tag}_open_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,total_ns,avg_ns,percent,min_ns,max_ns,calls_per_opening,ns_per_opening\n")
    for r in rows:
        k = short(r["Name"])
        f.write(f"{k},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{r['MinNs']},"
                f"{r['MaxNs']},{int(r['Calls']) / n_open:.2f},{float(r['TotalDurationNs']) / n_open:.0f}\n")
ns_per_open = sum(float(r["TotalDurationNs"]) for r in rows if short(r["Name"]) in POLY) / n_open

tot = collections.defaultdict(lambda: collections.defaultdict(float))      # counter -> kernel -> sum over the run
for sub in ("pmc_fetch", "pmc_write", "pmc_valu"):
    path = one(sub + "/**/*counter_collection.csv")
    if not path:
        continue
    for row in csv.DictReader(open(path)):
        tot[row["Counter_Name"]][short(row["Kernel_Name"])] += float(row["Counter_Value"])
counters = sorted(tot)
kernels = sorted({k for c in tot.values() for k in c})
with open(os.path.join(dst, f"{tag}_open_pmc.csv"), "w") as f:
    f.write("kernel," + ",".join(f"{c}_per_opening" for c in counters)
            + ",hbm_read_bytes_per_opening(2x FETCH KiB),hbm_write_bytes_per_opening,hbm_bytes_per_opening\n")
    for k in kernels:
        fk, wk = tot.get("FETCH_SIZE", {}).get(k, 0.0) / n_open, tot.get("WRITE_SIZE", {}).get(k, 0.0) / n_open
        f.write(k + "," + ",".join(f"{tot[c].get(k, 0.0) / n_open:.6g}" for c in counters)
                + f",{2 * fk * 1024:.0f},{wk * 1024:.0f},{(2 * fk + wk) * 1024:.0f}\n")
hbm = sum((2 * tot.get("FETCH_SIZE", {}).get(k, 0.0) + tot.get("WRITE_SIZE", {}).get(k, 0.0)) * 1024 for k in POLY) / n_open
valu = sum(tot.get("SQ_INSTS_VALU", {}).get(k, 0.0) for k in POLY) / n_open
cj_path = os.path.join(dst, "counters.json")
cj = json.load(open(cj_path)) if os.path.exists(cj_path) else {}
cj["open_poly"] = {"hbm_bytes": hbm, "valu_wave_instructions": valu, "ns_per_opening_under_profiler": ns_per_open,
                   "openings_profiled": n_open, "k": 6, "log_n": 20,
                   "source": f"profiles/{tag}_open_pmc.csv, profiles/{tag}_open_kernel_stats.csv (tools/profile_open.sh: "
                             "tools/open_only.py 20 6 12 under rocprofv3, one counter group per run)"}
json.dump(cj, open(cj_path, "w"), indent=1)
print(json.dumps(cj["open_poly"], indent=1))
