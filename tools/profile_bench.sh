#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py's
# headline loop, and the NTT alone.  Usage: tools/profile_bench.sh <tag>   -> gpurun_out/prof_<tag>/
# Counter passes are their own runs with --kernel-trace only (no sys/hip/hsa trace domains).
set -e
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
BENCH="python3 bench.py --mode batch --no-cpu-baseline --no-isolated"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH --steps 5 --warmup 2 > $OUT/trace_bench.log 2>&1
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $BENCH --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $BENCH --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/pmc_valu -o valu -- $BENCH --steps 3 --warmup 1 > $OUT/pmc_valu.log 2>&1
echo "valu done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_mem -o mem -- $BENCH --steps 3 --warmup 1 > $OUT/pmc_mem.log 2>&1 || echo "mem-instruction pass failed (optional)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ntt_alone -o ntt -- python3 tools/ntt_only.py 20 4 20 > $OUT/ntt_alone.log 2>&1
echo "ntt done"
find $OUT -name "*.csv" | head -40
