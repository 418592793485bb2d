#!/bin/bash
# One rocprofv3 --pmc pass of bench.py (headline loop only) and the per-kernel averages of the counters.
#   tools/pmc_pass.sh <out-name> "<COUNTER ...>" [library.so]      -> gpurun_out/pmc_<out-name>.txt
set -e
NAME=$1; CTRS=$2; LIB=$3
OUT=$PWD/gpurun_out/pmc_$NAME
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$LIB" ] && export KZG_MI355X_LIB=$LIB
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o p -- python3 bench.py --mode batch --steps 3 --warmup 1 --no-cpu-baseline --no-isolated > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY' | tee $OUT.txt
import collections, csv, glob, re, sys
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
    k = "rocprim" if "rocprim" in row["Kernel_Name"] else (m.group(1) if m else row["Kernel_Name"][:40])
    agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    print(f"{k:28s}", "  ".join(f"{c}={sum(v)/len(v):.4g} (x{len(v)})" for c, v in sorted(agg[k].items())))
PY
