#!/usr/bin/env python3
"""Where the 16x16x128 forward scan starts to pay: uniform and ragged databases around the threshold of the library's
estimate (Mfma16Cost, vsmatch.hip), all pairs, the library's choice against both kernels forced (knob bits 24 / 19)."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
vsm = importlib.import_module("visual-slam_amd")
synth = vsm.synth
dev = torch.device("cuda", 0)
M16, M16_OFF, TILE256 = 0x1000000, 0x80000, 0x40000000
rng = np.random.default_rng(5)
cases = [("uniform %d" % n, [n] * 128) for n in (800, 1000, 1200, 1400, 1600, 2000, 2500)]
cases += [("ragged 1200-1500", list(rng.integers(1200, 1501, size=128))), ("ragged 600-2400", list(rng.integers(600, 2401, size=128))),
          ("ragged 117-1363 (euroc_V1-like)", list(rng.integers(117, 1364, size=200)))]
for name, sizes in cases:
    n_img = len(sizes)
    db, _ = synth.sfm_database(n_img, max(sizes), 7, share_frac=0.3, window=8)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    parts = [db[i * max(sizes): i * max(sizes) + sizes[i]] for i in range(n_img)]
    db = np.ascontiguousarray(np.concatenate(parts))
    pairs = synth.all_pairs(n_img)
    with vsm.Matcher(0) as m:
        m.db_upload(db, off)
        m.set_profiling(True)
        dp = torch.from_numpy(pairs.view(np.int32).copy()).to(dev)
        doffs = torch.empty(len(pairs) + 1, dtype=torch.int64, device=dev)
        cap = len(pairs) * 400
        dout = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        res = {}
        for label, variant in (("choice", 0), ("16x16", M16 | TILE256), ("32x32", M16_OFF)):
            m.set_variant(variant)
            best = 1e9
            for _ in range(4):
                torch.cuda.synchronize()
                t = time.perf_counter()
                tot = m.match_pairs_device(dp.data_ptr(), len(pairs), 70, 1.2, doffs.data_ptr(), dout.data_ptr(), cap)
                best = min(best, time.perf_counter() - t)
            res[label] = (best, m.last_scan_kernel(0), m.last_scan_stats(0)["scan_ms"], tot)
        assert len({r[3] for r in res.values()}) == 1
        print(f"{name:34s} choice: kernel {res['choice'][1]} {res['choice'][0]*1e3:7.3f} ms | 16x16 {res['16x16'][0]*1e3:7.3f} ms (fwd {res['16x16'][2]:6.3f}) | "
              f"32x32 {res['32x32'][0]*1e3:7.3f} ms (fwd {res['32x32'][2]:6.3f})", flush=True)
