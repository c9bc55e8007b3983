set -e
for i in 1 2 3; do
  for v in fit nofit; do
    cp tools/bin/libvsmatch_$v.so visual-slam_amd/libvsmatch.so
    timeout -k 10 200 python bench.py --no-extra --cpu-pairs 0 --steps 30 --warmup 8 > gpurun_out/ab_${v}_$i.json 2>/dev/null
    python -c "
import json
d=json.loads(open('gpurun_out/ab_${v}_$i.json').readline())
r=d['roofline']; print('$v', $i, round(d['pairs_per_s']), round(r['avg_launch_ms'],4), round(r['kernel_gcomps']))"
  done
done
