#!/usr/bin/env python3
"""Sweeps (tile size, train splits) of the matrix-core scan on the 1M x 1M one-direction config."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
vsm = importlib.import_module("visual-slam_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1)
q = torch.randint(-2**63, 2**63 - 1, (n, 4), dtype=torch.int64, generator=g).to(dev)
t = torch.randint(-2**63, 2**63 - 1, (n, 4), dtype=torch.int64, generator=g).to(dev)
o = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3)]
m = vsm.Matcher(0)
m.set_profiling(True)
ref = None
for R, tile in ((8 | 0x40000000 | 0x80000, "256 (32x32x64)"), (8 | 0x40000000 | 0x1000000, "224 (16x16x128)"), (8 | 0x20000000, "128 (32x32x64)")):
    for splits in (0, 32, 40, 48, 64, 96, 128):
        m.set_variant(R | (splits << 8))
        best = 1e9
        for rep in range(3):
            m.nn2_device(q.data_ptr(), n, t.data_ptr(), n, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
            torch.cuda.synchronize()
            best = min(best, m.last_scan_stats()["scan_ms"])
        chk = (int(o[0].sum()), int(o[1].sum()), int(o[2].sum()))
        ref = ref or chk
        print(f"tile={tile:16s} splits={splits:3d} (0 = the library's choice)  {best:8.2f} ms  {n*n/best/1e6:8.1f} Gcomp/s  frac {n*n*512/best/1e9/10066.3:.3f}  {'ok' if chk == ref else 'MISMATCH'}", flush=True)
