#!/usr/bin/env python3
"""The front end over a whole image set (vsm_detect_and_describe_batch, sfm.cpp:1191-1201) against the
image-by-image loop: python3 tools/detect_batch_probe.py [n_img=164] [threads...]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_orb_oracle import synth_image  # noqa: E402

vsm = importlib.import_module("visual-slam_amd")
synth = importlib.import_module("visual-slam_amd.synth")


def timed(f, reps=5):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return min(t) * 1e3, sorted(t)[len(t) // 2] * 1e3


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 164
    # the 8 real EuRoC frames that travel with the tests, shifted (real frames give ~800 candidates and
    # ~300 keypoints; synth_image's texture gives 20 000 and 1300 -- pass "synth" as the last argument)
    if sys.argv[-1] == "synth":
        sys.argv.pop()
        imgs = [synth_image(752, 480, 100 + k) for k in range(n_img)]
    else:
        from tests.test_euroc_subset import image
        base = [image(k) for k in range(8)]
        imgs = [np.ascontiguousarray(np.roll(base[k % 8], (3 * (k // 8), 5 * (k // 8)), (0, 1))) for k in range(n_img)]
    with vsm.Matcher(0) as m:
        loop = timed(lambda: [m.detect_and_describe(im, 1500, True) for im in imgs], 3)
        print(f"image by image: {loop[0]:.2f} ms min, {loop[1]:.2f} ms median for {n_img} images ({loop[1] / n_img * 1e3:.0f} us per image)")
        for th in [int(a) for a in sys.argv[2:]] or [8]:
            os.environ["VSM_HOST_THREADS"] = str(th)
            b = timed(lambda: m.detect_and_describe_batch(imgs, 1500, True))
            k = timed(lambda: m.detect_and_describe_batch(imgs, 1500, True, keep_as_database=True, want=()))
            print(f"batch, {th} host threads: {b[0]:.2f} / {b[1]:.2f} ms ({b[1] / n_img * 1e3:.0f} us per image); into the database only: "
                  f"{k[0]:.2f} / {k[1]:.2f} ms")
        off, xy, ang, desc = m.detect_and_describe_batch(imgs, 1500, True, keep_as_database=True)
        pairs = synth.all_pairs(n_img)
        mt = timed(lambda: m.match_pairs(pairs, 70, 1.2))
        print(f"{off[-1]} keypoints; match_pairs of all {len(pairs)} pairs on the kept database: {mt[1]:.2f} ms")


if __name__ == "__main__":
    main()
