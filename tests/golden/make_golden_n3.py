#!/usr/bin/env python3
"""Writes tests/golden/ex4_corners.cereal: the `feature_corners` section of the reference's binary map
test/ex4_test_data/map_localize_camera.cereal (frames 0-9 of data/euroc_V1, both cameras: 20 images,
9708 corners with angles and descriptors) as a `corners.cereal` archive -- the file `sfm` writes after
detection (src/sfm.cpp:1203-1211) and reads back on the next start (959-1004).  Data only (the
reference's detector output in the reference's own on-disk format); runs where /root/reference exists.
The writer is the one tests/test_cereal_io.py pins byte for byte against the same map file."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
cio = importlib.import_module("visual-slam_amd.cereal_io")
src = "/root/reference/test/ex4_test_data/map_localize_camera.cereal"
m = cio.load_map_binary(src)
raw = open(src, "rb").read()
assert cio.dump_map_binary(m) == raw, "the writer must reproduce the reference's file byte for byte"
dst = os.path.join(ROOT, "tests", "golden", "ex4_corners.cereal")
cio.save_corners_binary(dst, m["corners"])
# a corners archive is the first section of the map archive: the bytes are the reference's own
assert raw.startswith(open(dst, "rb").read())
print(dst, os.path.getsize(dst), "bytes,", len(m["corners"]), "images,",
      sum(len(kd["corner_descriptors"]) for _, kd in m["corners"]), "corners")
