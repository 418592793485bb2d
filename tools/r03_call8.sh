#!/bin/bash
# round 3, GPU call 8: four gloo ranks on one GPU (rehearsal of the N = 4 line), full suite, final profile + default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python bench.py --gpus 4 --one-device --backend gloo --plonk-log-n 16 --range-log-n 18 --steps 5 --warmup 2 > gpurun_out/r03_bench_gloo4.json 2> gpurun_out/r03_bench_gloo4.err; echo "bench gloo4 rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_gloo4.json").read().strip().splitlines()[-1])
print(d["value"], d["collectives"], d["verified"], d["range_mode"]["verified"], d["distributed_ntt"]["forward_natural"]["verified"], d["plonk_round"]["verified"], d["plonk_round"]["value"])
PY
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r03_call8_pytest_full.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_call8_pytest_full.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
bash tools/profile_bench.sh r03 > gpurun_out/r03_profile_bench.log 2>&1; echo "profile rc=$?"
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
python tools/skew_check.py 2>&1 | grep "per commit"
