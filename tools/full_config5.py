"""BASELINE config 5 in full on one MI355X: 5000 images x 2000 descriptors, ALL 12 497 500
unordered pairs with cross-check, one vsm_match_pairs_device call (device-resident lists).
Checks a random sample of pairs against the CPU reference.  GPU box."""
import importlib, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vsm = importlib.import_module("visual-slam_amd")
from oracle import binding
synth = vsm.synth
n_img, n_desc = 5000, 2000
t0 = time.perf_counter()
db, off = synth.sfm_database(n_img, n_desc, 2025, share_frac=0.4, window=8)
pairs = synth.all_pairs(n_img)
print(f"database {db.nbytes / 1e6:.0f} MB, {len(pairs)} pairs, generated in {time.perf_counter() - t0:.1f} s", flush=True)
dev = torch.device("cuda", 0)
m = vsm.Matcher(0)
d_db = torch.from_numpy(db.view(np.int64)).to(dev)
m.db_attach(d_db.data_ptr(), off)
d_pairs = torch.from_numpy(pairs.view(np.int32).copy()).to(dev)
d_offs = torch.empty(len(pairs) + 1, dtype=torch.int64, device=dev)
cap = 80_000_000
d_out = torch.empty((cap, 2), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
t = time.perf_counter()
total = m.match_pairs_device(d_pairs.data_ptr(), len(pairs), 70, 1.2, d_offs.data_ptr(), d_out.data_ptr(), cap)
torch.cuda.synchronize()
dt = time.perf_counter() - t
offs = d_offs.cpu().numpy()
rng = np.random.default_rng(1)
sample = np.concatenate([rng.integers(0, len(pairs), 600), np.flatnonzero((pairs[:, 0] - pairs[:, 1]) <= 2)[:400]])
impl = binding.ref() or binding.port()
ok = True
for k in sample:
    i, j = pairs[k]
    want = impl.match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2)
    got = d_out[int(offs[k]):int(offs[k + 1])].cpu().numpy()
    ok &= bool(np.array_equal(got, want))
res = {"pairs": int(len(pairs)), "seconds": dt, "pairs_per_s": len(pairs) / dt, "matches": int(total),
       "reference_equivalent_gcomps": len(pairs) * 2.0 * n_desc * n_desc / dt / 1e9,
       "sample_checked_against_cpu_" + impl.kind: int(len(sample)), "sample_bit_exact": ok,
       "offsets_monotone": bool(np.all(np.diff(offs) >= 0)), "offsets_last_equals_total": bool(int(offs[-1]) == total)}
print(json.dumps(res))
