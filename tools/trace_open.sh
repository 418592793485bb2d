#!/bin/bash
# kernel timeline of pipelined openings (tools/open_only.py .. async): which kernels of the polynomial stage run beside
# which accumulate kernel.   tools/trace_open.sh <tag>  -> gpurun_out/trace_open_<tag>/timeline.txt
set -e
TAG=${1:-t}
OUT=$(pwd)/gpurun_out/trace_open_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/open_only.py 20 6 24 async > $OUT/unprofiled.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python3 tools/open_only.py 20 6 24 async > $OUT/run.log 2>&1
python3 tools/show_trace.py $OUT/trace_kernel_trace.csv 9.5 > $OUT/timeline.txt
cat $OUT/unprofiled.log | tail -1; tail -1 $OUT/run.log; wc -l $OUT/timeline.txt
