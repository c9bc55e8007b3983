#!/usr/bin/env python3
"""Backward-pass shape sweep on a dense-overlap batch (every pair shares landmarks)."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
vsm = importlib.import_module("visual-slam_amd")
synth = vsm.synth
dev = torch.device("cuda", 0)
N = 2000
ddb, doff = synth.sfm_database(96, N, 2028, share_frac=0.6, window=400)
dpairs = synth.all_pairs(96)
md = vsm.Matcher(0)
md.db_upload(ddb, doff)
dp = torch.from_numpy(dpairs.view(np.int32).copy()).to(dev)
doffs = torch.empty(len(dpairs) + 1, dtype=torch.int64, device=dev)
dout = torch.empty((len(dpairs) * N, 2), dtype=torch.int32, device=dev)
md.set_profiling(True)
ref = None
for name, variant in [("pruned bwdR=1", 1 << 20), ("pruned bwdR=2", 2 << 20), ("pruned bwdR=4", 4 << 20), ("pruned bwdR=8", 8 << 20), ("unpruned", 0x10000)]:
    md.set_variant(variant)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        tot = md.match_pairs_device(dp.data_ptr(), len(dpairs), 70, 1.2, doffs.data_ptr(), dout.data_ptr(), len(dpairs) * N)
        t = time.perf_counter() - t
        if best is None or t < best[0]:
            best = (t, md.last_scan_stats(0), md.last_scan_stats(1))
    chk = (tot, int(doffs.sum()), int(dout[:tot].to(torch.int64).sum()))
    ref = ref or chk
    t, f, b = best
    print(f"{name:16s} total {t*1e3:7.2f} ms  fwd {f['scan_ms']:6.2f} ms ({f['comparisons']/max(f['scan_ms'],1e-9)/1e6:6.0f} G/s)  "
          f"bwd {b['scan_ms']:6.2f} ms ({b['comparisons']/max(b['scan_ms'],1e-9)/1e6:6.0f} G/s)  pairs/s {len(dpairs)/t:9.0f}  {'ok' if chk == ref else 'MISMATCH'}", flush=True)
