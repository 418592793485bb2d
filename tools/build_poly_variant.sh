#!/bin/bash
# Build a variant of libkzg_mi355x.so that differs from the in-tree one only in poly.hip's compile flags
# (A/B timing of the opening's kernels):  tools/build_poly_variant.sh <name> [-DKZG_...=..]  -> ab/<name>/libkzg_mi355x.so
# (the in-tree objects of the other translation units are linked as they are; run `python -m kzg_snark_amd.build` first)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/ab/$NAME
mkdir -p $OUT
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-pass-failed"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $ROOT/kzg_snark_amd/csrc/poly.hip -o $OUT/poly.o
L=$ROOT/kzg_snark_amd/lib
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $L/api.o $L/ntt.o $L/msm.o $L/msm_prep.o $OUT/poly.o -o $OUT/libkzg_mi355x.so
echo $OUT/libkzg_mi355x.so
