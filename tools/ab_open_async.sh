#!/bin/bash
# Same-box alternating A/B of PIPELINED openings (tools/open_only.py 20 6 40 async: opens/s and the open_poly span beside
# the MSMs).  A variant is  name[:ENV=val[:ENV=val]]  with name = "tree" or a directory under ab/:
#   tools/ab_open_async.sh <rounds> tree tree:KZG_ACC_WGS_PER_CU=3 prio3 ...
R=$1; shift
for r in $(seq 1 $R); do
  for spec in "$@"; do
    IFS=: read -r -a parts <<< "$spec"
    v=${parts[0]}
    envs=("${parts[@]:1}")
    if [ "$v" = tree ]; then lib=""; else lib="KZG_MI355X_LIB=$PWD/ab/$v/libkzg_mi355x.so"; fi
    echo "$spec: $(env $lib "${envs[@]}" python tools/open_only.py 20 6 40 async 2>/dev/null | tail -1)"
  done
done
