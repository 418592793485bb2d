#!/bin/bash
# round 3, GPU call 7: fused combination + bottom evaluation level, LDS-staged final fill -- parity, A/B, profile
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kzg_gpu.py tests/test_golden_gpu.py tests/test_config4_gpu.py tests/test_plonk_oracle.py tests/test_vec_gpu.py -m gpu -x -q > gpurun_out/r03_call7_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_call7_pytest.log
for r in 1 2 3; do
  KZG_MI355X_LIB=$PWD/ab/head/libkzg_mi355x.so python tools/open_only.py 20 6 30 2>&1 | tail -1 | sed 's/^/head_r02:    /'
  KZG_MI355X_LIB=$PWD/ab/sweep2/libkzg_mi355x.so python tools/open_only.py 20 6 30 2>&1 | tail -1 | sed 's/^/dot:         /'
  python tools/open_only.py 20 6 30 2>&1 | tail -1 | sed 's/^/fused+staged: /'
done | tee gpurun_out/r03_open_ab2.txt
bash tools/profile_open.sh r03 > gpurun_out/r03_profile_open.log 2>&1; echo "profile_open rc=$?"
python bench.py --mode all --no-range --no-plonk --no-cpu-baseline --steps 10 > gpurun_out/r03_bench_open.json 2> gpurun_out/r03_bench_open.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_open.json").read().strip().splitlines()[-1])
print(d["value"], d["open"]["value"], d["open"]["pipelined"]["value"], d["open"]["poly_stage_ms"], d["open"]["verified"])
PY
