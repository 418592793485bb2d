#!/bin/bash
# kernel timeline of a short bench run (overlap between the pipeline's streams): tools/show_trace.py prints it
set -e
TAG=${1:-t}
shift || true
OUT=$(pwd)/gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python3 bench.py --mode batch --steps 6 --warmup 2 --no-cpu-baseline --no-isolated "$@" > $OUT/bench.log 2>&1
python3 tools/show_trace.py $OUT/trace_kernel_trace.csv 12.5 > $OUT/timeline.txt
wc -l $OUT/timeline.txt
