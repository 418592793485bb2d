#!/bin/bash
# kernel timeline of a short bench run (overlap between the pipeline's streams)
set -e
TAG=${1:-t}
OUT=$(pwd)/gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1
python3 - <<PY
import csv, re
rows=list(csv.DictReader(open("$OUT/trace_kernel_trace.csv")))
def short(n):
    m=re.search(r"(\w+_kernel)",n)
    return m.group(1) if m else ("sort" if "radix" in n or "rocprim" in n else n[:30])
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
# print the last ~70 kernels (steady state)
for r in rows[-260:]:
    print("%10.3f %10.3f  q%-3s %s"%((int(r["Start_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-t0)/1e6,r["Queue_Id"],short(r["Kernel_Name"])))
PY
