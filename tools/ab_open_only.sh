#!/bin/bash
# Same-box alternating A/B of the opening's polynomial stage alone (tools/open_only.py: HIP events around the
# "open_poly" span, k = 6, 2^20) for library variants under ab/ ("tree" = the in-tree library):
#   tools/ab_open_only.sh <rounds> <variant> [<variant> ...]
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ "$v" = tree ]; then unset KZG_MI355X_LIB; else export KZG_MI355X_LIB=$PWD/ab/$v/libkzg_mi355x.so; fi
    echo "$v: $(python tools/open_only.py 20 6 14 2>/dev/null | tail -1)"
  done
done
