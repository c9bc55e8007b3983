#!/usr/bin/env python3
"""Times the BoW database path on the GPU against the oracle's CPU loop: upload, the pair
selection of match_bow for a whole image set, single queries.  usage: bowdb_probe.py [n_img] [words/img] [vocab]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vsm = importlib.import_module("visual-slam_amd")
synth = importlib.import_module("visual-slam_amd.synth")
from oracle import binding  # noqa: E402  (a probe, not the product)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
vocab = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
t = time.perf_counter()
csr = synth.bow_vectors(n, vocab, per, seed=3, window=8)
print(f"{n} images x ~{per} words of {vocab}: {len(csr[1])} entries, generated in {time.perf_counter() - t:.1f} s", flush=True)
fr = np.arange(n) // 2
with vsm.Matcher(0) as m:
    for rep in range(3):
        t0 = time.perf_counter()
        m.bowdb_upload(csr)
        t1 = time.perf_counter()
        pairs = m.bow_select_pairs(fr, 25)
        t2 = time.perf_counter()
        print(f"upload {1e3 * (t1 - t0):.2f} ms, select_pairs {1e3 * (t2 - t1):.2f} ms, {len(pairs)} pairs", flush=True)
    off, w, v = csr
    q = [list(zip(w[off[i]:off[i + 1]].tolist(), v[off[i]:off[i + 1]].tolist())) for i in (0, n // 2, n - 1)]
    t0 = time.perf_counter()
    for _ in range(20):
        m.bowdb_query(q[:1], 25)
    print(f"one query: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms", flush=True)
t0 = time.perf_counter()
want, score = binding.match_bow_pairs([(int(f), i % 2) for i, f in enumerate(fr)], *csr, 25)
t1 = time.perf_counter()
a, b = set(map(tuple, want.tolist())), set(map(tuple, pairs.tolist()))
print(f"oracle loop (1 thread): {1e3 * (t1 - t0):.1f} ms, {len(want)} pairs; lists equal in order: "
      f"{np.array_equal(want.astype(np.int64), pairs.astype(np.int64))}, as sets: {a == b} ({len(a ^ b)} differ), "
      f"every score distinct: {len(set(score.tolist())) == len(score)}")
