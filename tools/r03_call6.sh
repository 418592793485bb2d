#!/bin/bash
# round 3, GPU call 6: partition-1 scatter experiments (bin-range sweeps, chunk sizes), RCCL rehearsal of the dealt proof
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_rccl_smoke_gpu.py tests/test_sharding_gloo.py tests/test_plonk.py -m gpu -x -q > gpurun_out/r03_call6_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_call6_pytest.log
for r in 1 2; do
  for v in base:kzg_snark_amd/lib sweep2:ab/sweep2 sweep4:ab/sweep4 sweep8:ab/sweep8 ch8192:ab/ch8192 ch16384:ab/ch16384; do
    name=${v%%:*}; lib=${v#*:}/libkzg_mi355x.so
    KZG_MI355X_LIB=$PWD/$lib python bench.py --mode batch --no-cpu-baseline --steps 30 > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || echo "FAILED $name"
    python - "$name" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab_{v}.json").read().strip().splitlines()[-1])
iso = d.get("kernel_ms_per_commit_isolated", {})
k = d["kernel_ms_per_commit"]
print(f"{v:10s} {d['value']:7.1f} commits/s  acc pipelined {k['msm_accumulate']:.3f} alone {d['roofline']['isolated']['avg_launch_ms']:.3f}  "
      f"pipelined p1 {k.get('msm_partition1',0):.3f} p2 {k.get('msm_partition2',0):.3f} | alone p1 {iso.get('msm_partition1',0):.3f} p2 {iso.get('msm_partition2',0):.3f}  ok {d['verified']['last_step_commit_trapdoor']}")
PY
  done
done | tee gpurun_out/r03_scatter1_ab.txt
for v in base:kzg_snark_amd/lib sweep4:ab/sweep4 ch16384:ab/ch16384; do
  name=${v%%:*}; lib=${v#*:}/libkzg_mi355x.so
  bash tools/pmc_pass.sh r03w_$name "WRITE_SIZE" $PWD/$lib > /dev/null 2>&1; echo "$name: $(grep 'prep_scatter1' gpurun_out/pmc_r03w_$name.txt)"
done | tee -a gpurun_out/r03_scatter1_ab.txt
python tools/skew_check.py > gpurun_out/r03_skew.txt 2>&1; tail -8 gpurun_out/r03_skew.txt
timeout -k 10 600 python bench.py --rehearse-collectives --steps 5 --warmup 2 --no-cpu-baseline --range-log-n 20 --plonk-log-n 16 > gpurun_out/r03_bench_rccl_one_rank.json 2> gpurun_out/r03_bench_rccl_one_rank.err; echo "rehearse rc=$?"; tail -c 700 gpurun_out/r03_bench_rccl_one_rank.json
