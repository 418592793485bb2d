#!/bin/bash
# round 3, GPU call 2: new GPU tests, open-stage profile, two-rank rehearsal of the bench line, 64-bit shift rates
set -o pipefail
mkdir -p gpurun_out
tools/microbench/int_rates > gpurun_out/r03_int_rates2.txt 2>&1 || echo "int_rates failed"
timeout -k 10 1500 python -m pytest tests/test_boundary_guards.py tests/test_sharding_gloo.py tests/test_kzg_gpu.py -m gpu -x -q > gpurun_out/r03_call2_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03_call2_pytest.log
bash tools/profile_open.sh r03 > gpurun_out/r03_profile_open.log 2>&1; echo "profile_open rc=$?"; tail -2 gpurun_out/r03_profile_open.log
timeout -k 10 900 python bench.py --gpus 2 --one-device --backend gloo --plonk-log-n 16 --range-log-n 18 --steps 5 --warmup 2 > gpurun_out/r03_bench_gloo2.json 2> gpurun_out/r03_bench_gloo2.err; echo "bench gloo2 rc=$?"; tail -c 1500 gpurun_out/r03_bench_gloo2.json
