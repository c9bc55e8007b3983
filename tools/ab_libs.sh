#!/bin/bash
# A/B of two builds of the library on one box: tools/bin/libvsmatch_a.so (the tree's) against
# tools/bin/libvsmatch_b.so, alternating; prints the headline, real-like and dense-overlap figures.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2 3; do
  for v in a b; do
    cp tools/bin/libvsmatch_$v.so visual-slam_amd/libvsmatch.so
    timeout -k 10 200 python bench.py --cpu-pairs 0 --steps 10 --warmup 5 > gpurun_out/ab_${v}_$i.json 2>/dev/null
    python -c "
import json
d=json.loads(open('gpurun_out/ab_${v}_$i.json').readline())
r=d['roofline']; rl=d['real_like']['matrix_core']; de=d['extra']['dense_overlap_96img']
print('$v', $i, 'headline', round(d['pairs_per_s']), 'real_like', round(rl['pairs_per_s']), round(rl['total_ms'],3), 'dense', round(de['pairs_per_s']), round(de['total_ms'],3))"
  done
done
