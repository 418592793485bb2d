#!/bin/bash
# round 3, GPU call 3: full GPU suite, open-stage A/B (dot-product lincomb) and its profile
set -o pipefail
mkdir -p gpurun_out
for r in 1 2; do
  KZG_MI355X_LIB=$PWD/ab/head/libkzg_mi355x.so python tools/open_only.py 20 6 30 2>&1 | tail -1 | sed 's/^/head_r02: /'
  python tools/open_only.py 20 6 30 2>&1 | tail -1 | sed 's/^/dot:      /'
done | tee gpurun_out/r03_open_ab.txt
bash tools/profile_open.sh r03 > gpurun_out/r03_profile_open.log 2>&1; echo "profile_open rc=$?"
timeout -k 10 1700 python -m pytest tests -m gpu -q > gpurun_out/r03_call3_pytest_full.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r03_call3_pytest_full.log
