"""Named parity cases shared by tests/golden/make_golden.py (which runs the real
reference on them, in the build container) and the tests (which compare the
oracle port and the HIP path with the recorded outputs).

A case is (d1 uint64[n1,4], d2 uint64[n2,4], threshold int, dist_2_best float).
"""
from __future__ import annotations

import importlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
synth = importlib.import_module("visual-slam_amd.synth")

ONES = np.uint64(0xFFFFFFFFFFFFFFFF)


def ex3():
    """The reference's own fixture: test/ex3_test_data/{kd0,kd1}.json
    (test_ex3.cpp:60-61) converted to packed words by make_golden.py."""
    return (np.load(os.path.join(GOLDEN, "ex3_kd0.npy")),
            np.load(os.path.join(GOLDEN, "ex3_kd1.npy")))


def _empty(n=0):
    return np.zeros((n, 4), dtype=np.uint64)


def _edge_cases():
    c = {}
    a, b = synth.orb_like_pair(40, 50, seed=7)
    c["edge_empty_both"] = (_empty(), _empty(), 70, 1.2)
    c["edge_empty_1"] = (_empty(), b, 70, 1.2)
    c["edge_empty_2"] = (a, _empty(), 70, 1.2)
    c["edge_1x1_same"] = (a[:1], a[:1].copy(), 70, 1.2)
    c["edge_1x1_far"] = (a[:1], a[:1] ^ ONES, 70, 1.2)  # distance 256
    c["edge_1xN"] = (a[:1], np.concatenate([b, a[:1]]), 70, 1.2)
    c["edge_Nx1"] = (np.concatenate([b, a[:1]]), a[:1], 70, 1.2)
    # all descriptors identical: best == second == 0, lowest index wins
    same = np.repeat(a[:1], 9, axis=0)
    c["edge_all_identical"] = (same, same[:5].copy(), 70, 1.2)
    c["edge_all_identical_ratio1"] = (same, same[:5].copy(), 70, 1.0)
    # exact duplicates of the best match inside the train set
    d2 = np.concatenate([b[:10], a[3:4], b[10:20], a[3:4], a[5:6]])
    c["edge_duplicate_best"] = (a[:8], d2, 70, 1.2)
    # complements only: every distance is 256 -> nothing recorded, idx -1
    c["edge_all_256"] = (same, same[:5] ^ ONES, 70, 1.2)
    c["edge_all_256_thr300_ratio1"] = (same, same[:5] ^ ONES, 300, 1.0)
    c["edge_all_256_thr300_ratio0"] = (same[:3], same[:2] ^ ONES, 300, 0.0)
    # threshold / ratio domain corners (sfm.cpp:200-202 ranges and beyond)
    for thr in (0, 1, 2, 255, 256, 257, 300, -5):
        c[f"edge_thr_{thr}"] = (a, b, thr, 1.2)
    for ratio in (0.0, 0.5, 1.0, 1.0000001, 2.0, 4.0, 300.0, -1.0):
        c[f"edge_ratio_{ratio}"] = (a, b, 70, ratio)
    c["edge_ratio_nan"] = (a, b, 70, float("nan"))
    c["edge_ratio_inf"] = (a, b, 70, float("inf"))
    c["edge_thr256_ratio1"] = (a, b, 256, 1.0)
    c["edge_thr300_ratio0"] = (a, b, 300, 0.0)
    # one bit apart / two candidates at equal distance
    base = a[:1].copy()
    near1 = base.copy(); near1[0, 0] ^= np.uint64(1)
    near2 = base.copy(); near2[0, 3] ^= np.uint64(1) << np.uint64(63)
    c["edge_tie_best"] = (base, np.concatenate([b[:4], near1, near2, b[4:8]]), 70, 1.0)
    c["edge_tie_best_ratio12"] = (base, np.concatenate([b[:4], near1, near2]), 70, 1.2)
    return c


def _ex3_cases():
    c = {}
    for thr in (60, 70, 80, 90, 100, 256):
        for ratio in (1.0, 1.1, 1.2, 1.5):
            c[f"ex3_thr{thr}_r{ratio}"] = ("ex3", thr, ratio)
    c["ex3_swapped"] = ("ex3_swapped", 70, 1.2)
    return c


# name -> (n1, n2, seed)
_PLANTED = {
    "planted_64x65": (64, 65, 11),
    "planted_257x1": (257, 1, 12),
    "planted_1x257": (1, 257, 13),
    "planted_300x777": (300, 777, 14),
    "planted_1023x1025": (1023, 1025, 15),
    "planted_1500x1500": (1500, 1500, 16),
    "planted_2000x2000": (2000, 2000, 17),
    "planted_537x547": (537, 547, 18),
    "planted_4100x130": (4100, 130, 19),
}


def case_names():
    return list(_ex3_cases()) + list(_edge_cases()) + list(_PLANTED)


def build_case(name):
    if name.startswith("ex3"):
        kind, thr, ratio = _ex3_cases()[name]
        d1, d2 = ex3()
        if kind == "ex3_swapped":
            d1, d2 = d2, d1
        return d1, d2, thr, ratio
    if name.startswith("edge"):
        return _edge_cases()[name]
    n1, n2, seed = _PLANTED[name]
    d1, d2 = synth.orb_like_pair(n1, n2, seed)
    return d1, d2, 70, 1.2


# small ragged database for the batched (match_all, sfm.cpp:1276-1353) path
SFM_SMALL = dict(n_img=7, sizes=[300, 0, 257, 264, 311, 1, 330], seed=21)


def sfm_small():
    db, off = synth.sfm_database(SFM_SMALL["n_img"], SFM_SMALL["sizes"], SFM_SMALL["seed"],
                                 share_frac=0.5, window=6)
    pairs = synth.all_pairs(SFM_SMALL["n_img"])
    return db, off, pairs
