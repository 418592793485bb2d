#!/bin/bash
# A/B of the open section (and the PLONK round, which leans on the vector primitives) for library variants under ab/
for v in "$@"; do
  if [ "$v" = tree ]; then unset KZG_MI355X_LIB; else export KZG_MI355X_LIB=$PWD/ab/$v/libkzg_mi355x.so; fi
  python bench.py --no-cpu-baseline --no-range --steps 10 > gpurun_out/abo_$v.json 2> gpurun_out/abo_$v.err || echo "FAILED $v"
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/abo_{v}.json").read().strip().splitlines()[-1])
o = d["open"]
print(f"{v:12s} open {o['value']:.1f}/s pipelined {o['pipelined']['value']:.1f}/s  {o['ms_per_open']:.3f} ms  poly stage {o['poly_stage_ms']*1e3:.1f} us  commits/s {d['value']:.1f}  ok {o['verified']}")
PY
  python tools/plonk_round.py --log-n 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('             plonk warm', d['prove_warm_s'], 'verified', d['verified'])"
done
