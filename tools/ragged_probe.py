"""Forward-scan throughput on the database with euroc_V1's real corner counts for several
kernel shapes (experiment knob of vsm_set_variant).  GPU box."""
import importlib, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vsm = importlib.import_module("visual-slam_amd")
synth = vsm.synth
rsz = np.array(json.load(open(os.path.join(ROOT, "tests/golden/euroc_subset/golden.json")))["euroc_V1_full_set"]["corners_per_image"])
dev = torch.device("cuda", 0)
sz3 = np.random.default_rng(3).integers(1200, 1501, size=164)
DATASETS = [
    ("euroc_V1 real sizes, 19800 pairs", synth.sfm_database(len(rsz), rsz, 2031, share_frac=0.5, window=12),
     synth.all_pairs(len(rsz), np.repeat(np.arange(len(rsz) // 2), 2))),
    ("164 img x 1200-1500, 13284 pairs", synth.sfm_database(164, sz3, 2029, share_frac=0.5, window=12),
     synth.all_pairs(164, np.repeat(np.arange(82), 2))),
    ("96 img x 2000 dense, 4560 pairs", synth.sfm_database(96, 2000, 2028, share_frac=0.6, window=400), synth.all_pairs(96)),
    ("400 img x 2000 sparse, 8192 pairs", synth.sfm_database(400, 2000, 2027, share_frac=0.4, window=8), synth.all_pairs(400)[:8192]),
    ("400 img x 2000 sparse, 2048 pairs", synth.sfm_database(400, 2000, 2027, share_frac=0.4, window=8), synth.all_pairs(400)[:2048]),
    ("200 img x 478 uniform, 19800 pairs", synth.sfm_database(200, 478, 2025, share_frac=0.5, window=12),
     synth.all_pairs(200, np.repeat(np.arange(100), 2))),
    ("200 img x 512 uniform, 19800 pairs", synth.sfm_database(200, 512, 2024, share_frac=0.5, window=12),
     synth.all_pairs(200, np.repeat(np.arange(100), 2))),
    ("40 img x 500, 780 pairs", synth.sfm_database(40, 500, 2026, share_frac=0.5, window=12), synth.all_pairs(40)),
]
ONLY_FIRST = len(sys.argv) > 1 and sys.argv[1] == 'first'  # profiling: the default shape on euroc_V1's sizes
for title, (db, off), pairs in (DATASETS[:1] if ONLY_FIRST else DATASETS):
    print(title)
    sizes = np.diff(off.astype(np.int64))
    m = vsm.Matcher(0)
    m.db_upload(db, off)
    rp = torch.from_numpy(pairs.view(np.int32).copy()).to(dev)
    offs = torch.empty(len(pairs) + 1, dtype=torch.int64, device=dev)
    cap = int(np.minimum(sizes[pairs[:, 0]], sizes[pairs[:, 1]]).sum())
    out = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    m.set_profiling(True)
    ref = None
    for name, v in ((("default", 0),) if ONLY_FIRST else (("default", 0), ("rmax 4", 0x40000008), ("rmax 2", 0x20000008), ("rmax 1", 0x10000008), ("latency shape", 1),
                                                                          ("rmax 4, 1 job/wave", 0x40100008), ("rmax 4, 2 jobs/wave", 0x40200008),
                                                                          ("rmax 4, 4 jobs/wave", 0x40400008), ("default again", 0))):
        m.set_variant(v)
        best = 1e9
        for _ in range(10):
            torch.cuda.synchronize(); t = time.perf_counter()
            tot = m.match_pairs_device(rp.data_ptr(), len(pairs), 70, 1.2, offs.data_ptr(), out.data_ptr(), cap)
            best = min(best, time.perf_counter() - t)
        k0, k1 = m.last_scan_stats(0), m.last_scan_stats(1)
        h = (tot, int(offs.sum().item()))
        ref = ref or h
        print(f"  {name:22s} total {best*1e3:7.3f} ms  {len(pairs)/best/1e6:6.3f} M pairs/s  fwd {k0['scan_ms']:6.3f} ms "
              f"{k0['comparisons']/max(k0['scan_ms'],1e-9)/1e6:7.1f} G/s  bwd {k1['scan_ms']:6.3f} ms  {'same' if h == ref else 'DIFF'}")
    m.close()
