#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/profile_bench.sh <tag>     outputs under gpurun_out/prof_<tag>/
set -e
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-isolated > $OUT/trace_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -30
