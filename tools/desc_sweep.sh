#!/bin/bash
# Forward-scan time per 32768-pair launch against descriptors per image (what a partial last step
# of the train range and the padding of the last query tile cost).  One box.
cd "${GRAFT_REPO_ROOT:-.}"
for n in 1984 2000 2016 2048 1792 1024; do
  timeout -k 10 200 python bench.py --no-extra --cpu-pairs 0 --steps 12 --warmup 3 --images 2000 --desc $n > gpurun_out/ds_$n.json 2>/dev/null
  python -c "
import json
d=json.loads(open('gpurun_out/ds_$n.json').readline())
r=d['roofline']; print($n, 'launch ms', round(r['avg_launch_ms'],4), 'kernel Gcomp/s', round(r['kernel_gcomps']), 'ns per step-block', round(r['avg_launch_ms']*1e6/32768/(($n+31)//32)/(($n+31)//32),3))"
done
