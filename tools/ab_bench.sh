#!/bin/bash
# A/B timing of library variants built with KZG_BUILD_DIR=ab/<name> (same box, back to back).
#   tools/ab_bench.sh name1 name2 ...    -> gpurun_out/ab_<name>.json
for v in "$@"; do
  if [ "$v" = tree ]; then unset KZG_MI355X_LIB; else export KZG_MI355X_LIB=$PWD/ab/$v/libkzg_mi355x.so; fi
  python bench.py --mode batch --no-cpu-baseline --steps 30 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || echo "FAILED $v"
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/ab_{v}.json").read().strip().splitlines()[-1])
    iso = d.get("kernel_ms_per_commit_isolated", {})
    print(f"{v:14s} {d['value']:7.1f} commits/s  acc pipelined {d['kernel_ms_per_commit']['msm_accumulate']:.3f} ms  alone {d['roofline']['isolated']['avg_launch_ms']:.3f} ms  ntt {d['ntt_ms']*1e3:.1f} us  "
          f"alone: p1 {iso.get('msm_partition1', 0):.3f} p2 {iso.get('msm_partition2', 0):.3f} order {iso.get('msm_order', 0):.3f} fin {iso.get('msm_finalize', 0):.3f} red {iso.get('msm_reduce', 0):.3f}  ok {d['verified']['last_step_commit_trapdoor']}")
except Exception as e:
    print(v, "no result", e)
PY
done
