#!/usr/bin/env python3
"""Regenerates tests/golden/euroc_subset/ (build container only; outputs are committed).

BASELINE config 3 ("SfM all-pairs matching on the euroc_V1 image set") on real frames, reduced
to what can travel to the GPU box: the first 4 stereo frames (8 images, data files of the
reference's data/euroc_V1/, byte for byte) and, in golden.json, what the CPU checkers compute
for them with the reference's call sequence (src/sfm.cpp:1191-1214, 1217-1273, 1276-1321):

  per image  detectKeypointsAndDescriptors(img, kd, 1500, true): corner count, sha256 of the
             corner list (float64 x, y) and of the packed descriptors   [C port, oracle/]
  per pair   matchDescriptors(kd_i, kd_j, 70, 1.2): count and sha256 of the int32 pair list,
             for the 4 stereo pairs (i,0)-(i,1) and the 24 non-stereo pairs j < i
             [the REAL reference function, oracle/_ref, cross-checked against the C port]
It also records, for DESIGN.md, the corner-count statistics of the whole euroc_V1 set.

Cross-pin: test/ex4_test_data/map_localize_camera.cereal (the reference's binary map, parsed
with visual-slam_amd/cereal_io.py) holds feature_corners of frames 0-9 of this very image set,
written by the reference's own detectKeypointsAndDescriptors.  This script asserts that the C
port reproduces all 20 of them (corner positions and order, descriptors bit for bit, angles to
1e-12) -- so the per-image hashes below are the reference's own outputs, not just the port's.
(The map's feature_matches are NOT outputs of this repository's matchDescriptors: 181 of the 190
lists are identical, the other 9 hold one extra pair each, always with first index 0 and a
failed backward ratio test -- a quirk of whatever implementation wrote that file.)
"""
import glob
import hashlib
import json
import os
import shutil
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
SRC = "/root/reference/data/euroc_V1"
DST = os.path.join(HERE, "euroc_subset")
N_FRAMES = 4


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    from oracle import binding
    port, ref = binding.port(), binding.ref()
    assert ref is not None, "build oracle/_ref first: make -C oracle ref"
    files = sorted(glob.glob(os.path.join(SRC, "*.jpg")))
    stamps = sorted({os.path.basename(f).split("_")[0] for f in files})
    sizes = [len(port.detect_keypoints(np.array(Image.open(f)), 1500)) for f in files]
    import importlib
    cio = importlib.import_module("visual-slam_amd.cereal_io")
    ref_map = dict(cio.load_map_binary("/root/reference/test/ex4_test_data/map_localize_camera.cereal")["corners"])
    for (fr, cam), kd in sorted(ref_map.items()):
        img = np.array(Image.open(os.path.join(SRC, f"{stamps[fr]}_{cam}.jpg")))
        xy = port.detect_keypoints(img, 1500)
        ang, desc = port.orb_describe(img, xy, True)
        assert np.array_equal(xy, kd["corners"]) and np.array_equal(desc, kd["corner_descriptors"]), (fr, cam)
        assert np.abs(ang - kd["corner_angles"]).max() < 1e-12
    print("C port reproduces feature_corners of the reference's map file for", len(ref_map), "images")
    os.makedirs(DST, exist_ok=True)
    names = [f"{s}_{c}.jpg" for s in stamps[:N_FRAMES] for c in (0, 1)]
    kds = []
    images = []
    for nm in names:
        shutil.copyfile(os.path.join(SRC, nm), os.path.join(DST, nm))
        os.chmod(os.path.join(DST, nm), 0o644)
        img = np.array(Image.open(os.path.join(DST, nm)))
        xy = port.detect_keypoints(img, 1500)
        ang, desc = port.orb_describe(img, xy, True)
        kds.append(desc)
        images.append({"file": nm, "corners": len(xy), "corners_sha256": sha(xy), "descriptors_sha256": sha(desc)})
    pairs = []
    frame = [k // 2 for k in range(len(names))]
    todo = [(2 * f, 2 * f + 1, "stereo") for f in range(N_FRAMES)]
    todo += [(i, j, "all") for i in range(len(names)) for j in range(i) if frame[i] != frame[j]]  # sfm.cpp:1285-1290
    for i, j, kind in todo:
        m = ref.match(kds[i], kds[j], 70, 1.2)
        assert np.array_equal(m, port.match(kds[i], kds[j], 70, 1.2))
        pairs.append({"i": i, "j": j, "kind": kind, "matches": len(m), "sha256": sha(m.astype(np.int32))})
    with open(os.path.join(DST, "golden.json"), "w") as f:
        json.dump({"images": images, "pairs": pairs, "images_verified_against_reference_map_file": len(ref_map),
                   "euroc_V1_full_set": {"images": len(files), "corners_min": int(min(sizes)),
                                         "corners_mean": float(np.mean(sizes)), "corners_max": int(max(sizes)),
                                         "non_stereo_pairs": len(files) * (len(files) - 1) // 2 - len(stamps),
                                         "corners_per_image": [int(v) for v in sizes]}},
                  f, indent=1)
    print(json.dumps(images, indent=1)[:400], "\n", len(pairs), "pairs", sum(p["matches"] for p in pairs), "matches")


if __name__ == "__main__":
    main()
