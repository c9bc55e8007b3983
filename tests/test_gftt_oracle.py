"""CPU checks of the corner-detector oracle (oracle/vsm_oracle_gftt.c = visnav::detectKeypoints,
keypoints.h:133-149 = cv::goodFeaturesToTrack + border filter) against the reference's golden
corner lists (kd{0,1}.json are detectKeypoints outputs on 0_{0,1}.jpg, test_ex3.cpp:76-83) and
against an independent numpy restatement of the published algorithm."""
import numpy as np
import pytest

from oracle import binding
from tests.test_orb_oracle import ex3_image, ex3_kd, synth_image

f32 = np.float32


def numpy_gftt(img, max_corners=1500, quality=0.01, min_dist=8.0):
    """cv::goodFeaturesToTrack restated with numpy float32 arrays (second, independent writing)."""
    h, w = img.shape
    p = np.pad(img.astype(f32), 1, mode="reflect")
    f1 = f32(1.0 / (4 * 3 * 255.0))
    f0 = f32(2) * f1
    rx = p[:, 2:] - p[:, :-2]
    dx = (rx[1:-1] * f0 + (rx[:-2] + rx[2:]) * f1).astype(f32)
    ry = (p[:, 1:-1] * f0 + (p[:, :-2] + p[:, 2:]) * f1).astype(f32)
    dy = (ry[2:] - ry[:-2]).astype(f32)

    def box(a):
        q = np.pad(a, 1, mode="reflect").astype(np.float64)
        rs = q[:, :-2] + q[:, 1:-1] + q[:, 2:]
        return (rs[:-2] + rs[1:-1] + rs[2:]).astype(f32)

    a, b, c = box(dx * dx) * f32(0.5), box(dx * dy), box(dy * dy) * f32(0.5)
    eig = ((a + c) - np.sqrt((a - c) * (a - c) + b * b)).astype(f32)
    thr = f32(np.float64(eig.max()) * quality)
    e = np.where(eig > thr, eig, f32(0))
    q = np.pad(e, 1, mode="constant", constant_values=-np.inf)
    dil = np.max([q[j:j + h, i:i + w] for j in range(3) for i in range(3)], axis=0)
    ys, xs = np.nonzero((e != 0) & (e == dil))
    keep = (ys >= 1) & (ys < h - 1) & (xs >= 1) & (xs < w - 1)
    ys, xs = ys[keep], xs[keep]
    order = np.lexsort((-(ys * w + xs), -e[ys, xs].astype(np.float64)))
    ys, xs = ys[order], xs[order]
    out = []
    if min_dist >= 1:
        cell = int(round(min_dist))
        gw, gh = (w + cell - 1) // cell, (h + cell - 1) // cell
        grid = [[[] for _ in range(gw)] for _ in range(gh)]
        for y, x in zip(ys, xs):
            xc, yc = x // cell, y // cell
            good = all((x - px) ** 2 + (y - py) ** 2 >= min_dist * min_dist
                       for yy in range(max(0, yc - 1), min(gh - 1, yc + 1) + 1)
                       for xx in range(max(0, xc - 1), min(gw - 1, xc + 1) + 1) for (px, py) in grid[yy][xx])
            if good:
                grid[yc][xc].append((x, y))
                out.append((x, y))
                if max_corners > 0 and len(out) == max_corners:
                    break
    else:
        out = list(zip(xs, ys))[: max_corners if max_corners > 0 else None]
    return np.array(out, dtype=np.float64).reshape(-1, 2), eig


GFTT_CASES = [  # (w, h, seed, max_corners, quality, min_distance)
    (96, 64, 1, 1500, 0.01, 8.0), (200, 150, 2, 50, 0.01, 8.0), (64, 48, 3, 0, 0.05, 3.0), (40, 40, 4, 1500, 0.01, 8.0),
    (33, 7, 5, 10, 0.2, 0.0), (3, 3, 6, 5, 0.01, 1.0), (5, 4, 7, 0, 0.001, 2.5), (130, 90, 8, 1500, 0.5, 20.0),
]


@pytest.mark.parametrize("k", [0, 1])
def test_oracle_reproduces_reference_corner_lists(k):
    """537 / 547 corners, same positions, same order (the order defines the FeatureIds that
    matches.json refers to)."""
    want = ex3_kd(k)[0]
    got = binding.port().detect_keypoints(ex3_image(k), 1500)
    assert got.shape == want.shape and np.array_equal(got, want)
    # num_features below the number found truncates the same ordered list, then filters
    few = binding.port().detect_keypoints(ex3_image(k), 100)
    assert len(few) <= 100 and np.array_equal(few, want[: len(few)])


@pytest.mark.parametrize("case", GFTT_CASES)
def test_oracle_equals_numpy_restatement(case):
    w, h, seed, mc, q, md = case
    img = synth_image(w, h, seed) if min(w, h) >= 8 else np.random.default_rng(seed).integers(0, 256, (h, w)).astype(np.uint8)
    pts, eig = numpy_gftt(img, mc, q, md)
    assert np.array_equal(binding.port().gftt_mineig(img), eig)
    assert np.array_equal(binding.port().gftt(img, mc, q, md), pts)


def test_flat_and_plateau_images():
    o = binding.port()
    assert len(o.gftt(np.full((40, 50), 77, np.uint8), 100)) == 0           # all eigenvalues 0
    step = np.zeros((40, 50), np.uint8)
    step[:, 25:] = 200                                                       # an edge: min eigenvalue 0 everywhere
    assert len(o.gftt(step, 100)) == 0
    yy, xx = np.indices((48, 64))
    chk = ((yy // 8 + xx // 8) % 2 * 255).astype(np.uint8)                  # checkerboard: many exactly equal corners (ties)
    pts, _ = numpy_gftt(chk, 0, 0.01, 4.0)
    assert len(pts) > 10 and np.array_equal(o.gftt(chk, 0, 0.01, 4.0), pts)
