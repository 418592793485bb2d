#!/bin/bash
# Runs on the GPU box (via gpurun): the opening's polynomial stage (csrc/poly.hip: tile_combine_kernel,
# tile_fill_kernel; kzg.py:148-154) under rocprofv3 -- kernel-trace stats, then FETCH_SIZE,
# WRITE_SIZE and SQ_INSTS_VALU in separate --pmc passes (kernel-trace only, the program directly after `--`).
#   tools/profile_open.sh <tag>   -> gpurun_out/prof_open_<tag>/ ; then: python tools/summarize_open.py <tag>
set -e
TAG=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_open_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
python3 tools/open_only.py 20 6 12 > $OUT/unprofiled.log 2>&1
RUN="python3 tools/open_only.py 20 6 12"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $RUN > $OUT/trace.log 2>&1
echo "open trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $RUN > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $RUN > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_valu -o valu -- $RUN > $OUT/pmc_valu.log 2>&1
echo "open pmc done"
cat $OUT/unprofiled.log | tail -1
