#!/usr/bin/env python3
"""Writes tests/golden/ex3_bowdb.npz: the data the reference's own tests hold for BowDatabase
(bow_db.h) and for the accumulation half of BowVocabulary::transform (bow_voc.h:107-130), none of
which needs the missing vocabulary file:
  test/ex3_test_data/bow_db.json      the inverted index after inserting 43 images (test_ex3.cpp:447-514)
  test/ex3_test_data/bow_dist.cereal  the L1 distance of every image to every image (test_ex3.cpp:579-653)
  test/ex3_test_data/bow_res.json     word id and weight of each of kd0's 537 descriptors (351-385)
  test/ex3_test_data/bow_res2.json    the BoW vectors of kd0 and kd1 (387-445)
Data only, flattened into arrays; runs where /root/reference exists."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
cio = importlib.import_module("visual-slam_amd.cereal_io")
R = "/root/reference/test/ex3_test_data/"
index = cio.load_bow_db_json(R + "bow_db.json")
fc, dist = cio.load_bow_dist_binary(R + "bow_dist.cereal")
n = dist.shape[0]
assert dist.shape == (n, n) and np.array_equal(fc[..., 0], np.tile(np.arange(n), (n, 1))) and not fc[..., 1].any()
res = json.load(open(R + "bow_res.json"))
res2 = json.load(open(R + "bow_res2.json"))
out = {
    "inv_word": np.array([w for w, lst in index for _ in lst], np.uint32),
    "inv_frame": np.array([f[0] for _, lst in index for f, _ in lst], np.int64),
    "inv_cam": np.array([f[1] for _, lst in index for f, _ in lst], np.int64),
    "inv_value": np.array([v for _, lst in index for _, v in lst], np.float64),
    "dist": dist,
    "kd0_word": np.array(res["w_id"], np.uint32), "kd0_weight": np.array(res["w_val"], np.float64),
}
for k in ("bow0", "bow1"):
    out[k + "_word"] = np.array([e["first"] for e in res2[k]], np.uint32)
    out[k + "_value"] = np.array([e["second"] for e in res2[k]], np.float64)
dst = os.path.join(ROOT, "tests", "golden", "ex3_bowdb.npz")
np.savez_compressed(dst, **out)
print(dst, os.path.getsize(dst), "bytes;", n, "images,", len(index), "words,", len(out["inv_word"]), "entries")
