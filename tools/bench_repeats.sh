#!/bin/bash
# Headline variance on one box and its dependence on the batch size: tools/bench_repeats.sh > gpurun_out/bench_repeats.txt
for rep in 1 2 3; do
  python bench.py --mode batch --no-cpu-baseline --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch 4  run $rep: %.1f commits/s  %.3f ms/step  ok %s' % (d['value'], d['ms_per_step'], d['verified']['last_step_commit_trapdoor']))"
done
for b in 1 8 16; do
  python bench.py --mode batch --no-cpu-baseline --no-isolated --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch %-2d        : %.1f commits/s  %.3f ms/step  ok %s' % ($b, d['value'], d['ms_per_step'], d['verified']['last_step_commit_trapdoor']))"
done
