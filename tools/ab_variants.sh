#!/bin/bash
# Headline batch under experiment-knob variants, one box, alternating.  usage: ab_variants.sh v1 v2 ...
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do
  for v in "$@"; do
    timeout -k 10 200 python bench.py --no-extra --cpu-pairs 0 --steps 20 --warmup 5 --variant $v > gpurun_out/var_${v}_$i.json 2>/dev/null
    python -c "
import json
d=json.loads(open('gpurun_out/var_${v}_$i.json').readline())
r=d['roofline']; print('variant', hex($v), 'rep', $i, 'pairs/s', round(d['pairs_per_s']), 'launches', r['launches'], 'launch ms', round(r['avg_launch_ms'],4), 'kernel G/s', round(r['kernel_gcomps']))"
  done
done
