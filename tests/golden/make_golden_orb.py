#!/usr/bin/env python3
"""Regenerates the ORB-extraction goldens from the reference's test data (build container
only; outputs are committed).  Pins of computeAngles / computeDescriptors
(keypoints.h:151-237; test/src/test_ex3.cpp:76-146):

  ex3_0_{0,1}.jpg            the reference's test images, byte for byte (data files)
  ex3_kd{0,1}_corners.npy    float64[n,2]  kd{0,1}.json "m.corners"
  ex3_kd{0,1}_angles.npy     float64[n]    kd{0,1}.json "m.corner_angles"
  (ex3_kd{0,1}.npy, the descriptors, are written by make_golden.py)
"""
import importlib
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference/test/ex3_test_data"


def main():
    cio = importlib.import_module("visual-slam_amd.cereal_io")
    for k in (0, 1):
        dst = os.path.join(HERE, f"ex3_0_{k}.jpg")
        shutil.copyfile(os.path.join(REF, f"0_{k}.jpg"), dst)
        os.chmod(dst, 0o644)
        kd = cio.load_keypoints_json(os.path.join(REF, f"kd{k}.json"))
        np.save(os.path.join(HERE, f"ex3_kd{k}_corners.npy"), kd["corners"])
        np.save(os.path.join(HERE, f"ex3_kd{k}_angles.npy"), kd["corner_angles"])
        assert np.array_equal(kd["corner_descriptors"], np.load(os.path.join(HERE, f"ex3_kd{k}.npy")))


if __name__ == "__main__":
    main()
