#!/bin/bash
set -e
OUT=$(pwd)/gpurun_out/prof_ntt
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/a -o a -- python3 tools/ntt_only.py 20 4 6 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o b -- python3 tools/ntt_only.py 20 4 6 > $OUT/b.log 2>&1 || true
python3 - <<PY
import csv, collections, glob
for sub in ("a","b"):
    fs=glob.glob("$OUT/%s/*counter_collection.csv"%sub)
    if not fs: print(sub,"no counters"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        if "ntt_pass" in row["Kernel_Name"]:
            agg[row["Counter_Name"]]["v"].append(float(row["Counter_Value"]))
    for k,v in agg.items():
        vals=v["v"]; print(sub,k,"avg per launch = %.4g over %d launches"%(sum(vals)/len(vals),len(vals)))
PY
