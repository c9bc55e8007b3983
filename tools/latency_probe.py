#!/usr/bin/env python3
"""Host-to-host latency of one stereo pair (config 2) through vsm_match_pair."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vsm = importlib.import_module("visual-slam_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
l, r = vsm.synth.orb_like_pair(n, n, 2026)
m = vsm.Matcher(0)
for variant in [0] + [int(v, 0) for v in sys.argv[2:]]:
    m.set_variant(variant)
    for _ in range(20):
        res = m.match(l, r, 70, 1.2)
    ts = []
    for _ in range(200):
        t = time.perf_counter(); res = m.match(l, r, 70, 1.2); ts.append(time.perf_counter() - t)
    ts.sort()
    print(f"variant {variant:#x}: n={n} matches={len(res)} median {ts[100]*1e6:.1f} us  p10 {ts[20]*1e6:.1f} us  min {ts[0]*1e6:.1f} us", flush=True)
