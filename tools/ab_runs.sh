#!/bin/bash
# Headline batch by consecutive jobs per wave of the forward scan (knob bits 20-23), one box, alternating.
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do
  for j in 1 2 4 8 3; do
    v=$((j << 20))
    timeout -k 10 200 python bench.py --no-extra --cpu-pairs 0 --steps 20 --warmup 5 --variant $v > gpurun_out/runs_${j}_$i.json 2>/dev/null
    python -c "
import json
d=json.loads(open('gpurun_out/runs_${j}_$i.json').readline())
r=d['roofline']; print('jobs per wave', $j, 'rep', $i, 'pairs/s', round(d['pairs_per_s']), 'launch ms', round(r['avg_launch_ms'],4), 'kernel G/s', round(r['kernel_gcomps']), 'frac', round(r['frac'],4))"
  done
done
