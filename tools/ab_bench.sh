#!/bin/bash
# A/B timing of library variants built with KZG_BUILD_DIR=ab/<name> (same box, back to back).
#   tools/ab_bench.sh name1 name2 ...    -> gpurun_out/ab_<name>.json
for v in "$@"; do
  KZG_MI355X_LIB=$PWD/ab/$v/libkzg_mi355x.so python bench.py --mode batch --no-cpu-baseline --steps 30 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || echo "FAILED $v"
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/ab_{v}.json").read().strip().splitlines()[-1])
    print(f"{v:14s} {d['value']:7.1f} commits/s  acc pipelined {d['kernel_ms_per_commit']['msm_accumulate']:.3f} ms  alone {d['roofline']['isolated']['avg_launch_ms']:.3f} ms  ntt {d['ntt_ms']*1e3:.1f} us  verified {d['verified']}")
except Exception as e:
    print(v, "no result", e)
PY
done
