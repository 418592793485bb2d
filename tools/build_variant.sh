#!/bin/bash
# Build a variant of libkzg_mi355x.so that differs from the in-tree one only in ONE translation unit's compile flags
# (A/B timing):  tools/build_variant.sh <unit: poly|msm_prep|ntt|msm|api> <name> [-DKZG_...=..]  -> ab/<name>/libkzg_mi355x.so
# (the in-tree objects of the other units are linked as they are; run `python -m kzg_snark_amd.build` first)
set -e
UNIT=$1; NAME=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/ab/$NAME
mkdir -p $OUT
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-pass-failed"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $ROOT/kzg_snark_amd/csrc/$UNIT.hip -o $OUT/$UNIT.o
L=$ROOT/kzg_snark_amd/lib
OBJS=""
for u in api ntt msm msm_prep poly; do
  if [ $u = $UNIT ]; then OBJS="$OBJS $OUT/$u.o"; else OBJS="$OBJS $L/$u.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -o $OUT/libkzg_mi355x.so
echo $OUT/libkzg_mi355x.so
