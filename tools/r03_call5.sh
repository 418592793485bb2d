#!/bin/bash
# round 3, GPU call 5: register-held bin sort parity + A/B, full rocprofv3 profile of the build, default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kzg_gpu.py tests/test_golden_gpu.py tests/test_config4_gpu.py -m gpu -x -q > gpurun_out/r03_call5_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r03_call5_pytest.log
for r in 1 2; do
  for v in head_r02:ab/head/libkzg_mi355x.so binsort_regs:kzg_snark_amd/lib/libkzg_mi355x.so; do
    name=${v%%:*}; lib=${v#*:}
    KZG_MI355X_LIB=$PWD/$lib python bench.py --mode batch --no-cpu-baseline --steps 30 > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || echo "FAILED $name"
    python - "$name" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab_{v}.json").read().strip().splitlines()[-1])
iso = d.get("kernel_ms_per_commit_isolated", {})
k = d["kernel_ms_per_commit"]
print(f"{v:12s} {d['value']:7.1f} commits/s  acc pipelined {k['msm_accumulate']:.3f} alone {d['roofline']['isolated']['avg_launch_ms']:.3f}  ntt {d['ntt_ms']*1e3:.1f} us  "
      f"pipelined p1 {k.get('msm_partition1',0):.3f} p2 {k.get('msm_partition2',0):.3f} | alone p1 {iso.get('msm_partition1',0):.3f} p2 {iso.get('msm_partition2',0):.3f} order {iso.get('msm_order',0):.3f} red {iso.get('msm_reduce',0):.3f}  ok {d['verified']['last_step_commit_trapdoor']}")
PY
  done
done | tee gpurun_out/r03_prep_ab2.txt
bash tools/profile_bench.sh r03 > gpurun_out/r03_profile_bench.log 2>&1; echo "profile rc=$?"
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r03_bench_default.json
