#!/bin/bash
# rocprofv3 kernel-trace stats of the NTT alone (a batch of 4 transforms of 2^20, 10 launches of each direction):
# the isolated counterpart of the ntt_pass_kernel row in the pipelined bench profile.
# Usage (on the GPU box): tools/profile_ntt_stats.sh <tag>  ->  gpurun_out/prof_<tag>_ntt/
set -e
TAG=${1:-r01}
OUT=$(pwd)/gpurun_out/prof_${TAG}_ntt
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ntt -- python3 tools/ntt_only.py 20 4 20 > $OUT/ntt.log 2>&1
cat $OUT/ntt_kernel_stats.csv
