"""Forward-scan rate on descriptors with the distance statistics of REAL ORB descriptors.

The synthetic databases of bench.py are iid Bernoulli(1/2) bits plus planted matches: two
unrelated descriptors are 128 +- 8 bits apart and practically never closer than the accept
limit (83 for threshold 70 / ratio 1.2).  Real ORB descriptors are correlated: in the
reference's ex3 stereo pair (tests/golden/ex3_kd*.npy) unrelated descriptors are 126 +- 19 apart
and 2.2 % of all pairs are closer than 84, so 95 % of the 32 x 32 blocks of the matrix-core
scan hold a candidate.  This probe builds a database from those 1084 real descriptors (each
image = a random subset, every descriptor with ~8 random bits flipped) and times the batched
matcher on it.  GPU box.   usage: real_like_probe.py [images] [descriptors per image]"""
import importlib, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vsm = importlib.import_module("visual-slam_amd")
synth = vsm.synth
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n_desc = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
real = np.concatenate([np.load(os.path.join(ROOT, "tests", "golden", "ex3_kd0.npy")),
                       np.load(os.path.join(ROOT, "tests", "golden", "ex3_kd1.npy"))])
rng = np.random.default_rng(7)
parts = []
for i in range(n_img):
    idx = rng.choice(len(real), size=n_desc, replace=n_desc > len(real))
    parts.append(real[idx] ^ synth.flip_masks(n_desc, 99, 500 + i, levels=(5,)))
db = np.ascontiguousarray(np.concatenate(parts))
off = (np.arange(n_img + 1) * n_desc).astype(np.uint32)
pairs = synth.all_pairs(n_img)
dev = torch.device("cuda", 0)
m = vsm.Matcher(0)
m.db_upload(db, off)
d_pairs = torch.from_numpy(pairs.view(np.int32).copy()).to(dev)
d_offs = torch.empty(len(pairs) + 1, dtype=torch.int64, device=dev)
cap = len(pairs) * n_desc
d_out = torch.empty((cap, 2), dtype=torch.int32, device=dev)
res = {"images": n_img, "desc_per_image": n_desc, "pairs": int(len(pairs))}
lists = {}
for name, variant in (("matrix_core", 0), ("popcount", 0x2000000), ("matrix_core_128_query_tiles", 0x20000000),
                      ("matrix_core_64_query_tiles", 0x10000000), ("matrix_core_unpruned", 0x10000), ("matrix_core_again", 0)):
    m.set_variant(variant)
    m.match_pairs_device(d_pairs.data_ptr(), len(pairs), 70, 1.2, d_offs.data_ptr(), d_out.data_ptr(), cap)
    m.set_profiling(True)
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    total = m.match_pairs_device(d_pairs.data_ptr(), len(pairs), 70, 1.2, d_offs.data_ptr(), d_out.data_ptr(), cap)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    f, b = m.last_scan_stats(0), m.last_scan_stats(1)
    m.set_profiling(False)
    lists[name] = (d_offs.cpu().numpy().copy(), d_out[:total].cpu().numpy().copy())
    res[name] = {"pairs_per_s": len(pairs) / dt, "total_ms": dt * 1e3, "matches_per_pair": total / len(pairs),
                 "forward_ms": f["scan_ms"], "forward_gcomps": f["comparisons"] / (f["scan_ms"] * 1e-3) / 1e9,
                 "backward_ms": b["scan_ms"], "backward_gcomps": b["comparisons"] / max(b["scan_ms"], 1e-9) / 1e6}
res["lists_identical"] = all(bool(np.array_equal(lists["popcount"][0], v[0]) and np.array_equal(lists["popcount"][1], v[1]))
                             for v in lists.values())
print(json.dumps(res))
