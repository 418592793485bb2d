#!/bin/bash
# round 3, GPU call 1: integer issue rates, NTT bisect (same box, alternating), new parity tests
set -o pipefail
mkdir -p gpurun_out
tools/microbench/int_rates > gpurun_out/r03_int_rates.txt 2>&1 || echo "int_rates failed"
python tools/ntt_ab.py --rounds 3 \
  r01h=ab/r01h/libkzg_mi355x.so \
  113110a_exchange_layouts=ab/113110a/libkzg_mi355x.so \
  5aae198_columnwise_mul=ab/5aae198/libkzg_mi355x.so \
  5285154_chain_pin=ab/5285154/libkzg_mi355x.so \
  7a6a423_factor_twist=ab/7a6a423/libkzg_mi355x.so \
  head_r02=ab/head/libkzg_mi355x.so \
  head_r02_twist_table=ab/head/libkzg_mi355x.so,KZG_NTT_TWIST_TABLE=1 \
  head_r02_nopin=ab/head_nopin/libkzg_mi355x.so \
  v3a=kzg_snark_amd/lib/libkzg_mi355x.so \
  v3a_twist_table=kzg_snark_amd/lib/libkzg_mi355x.so,KZG_NTT_TWIST_TABLE=1 \
  v3a_nopin=ab/v3a_nopin/libkzg_mi355x.so \
  > gpurun_out/r03_ntt_bisect.txt 2>&1
tail -14 gpurun_out/r03_ntt_bisect.txt
timeout -k 10 900 python -m pytest tests/test_ntt_gpu.py tests/test_golden_gpu.py -m gpu -x -q > gpurun_out/r03_call1_pytest_ntt.log 2>&1; echo "pytest ntt rc=$?"; tail -3 gpurun_out/r03_call1_pytest_ntt.log
