"""Latency breakdown of the detection / description entry points on one frame (GPU box)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vsm = importlib.import_module("visual-slam_amd")
from PIL import Image
rng = np.random.default_rng(2025)
tex = rng.integers(0, 256, size=(480 // 6 + 2, 752 // 6 + 2)).astype(np.float64)
timg = np.clip(np.kron(tex, np.ones((6, 6)))[:480, :752] * 0.8 + rng.integers(0, 50, (480, 752)), 0, 255).astype(np.uint8)
real = np.array(Image.open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ex3_0_0.jpg")))
m = vsm.Matcher(0)
def t(f, reps=30):
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    return (time.perf_counter() - t0) / reps * 1e3, r
for name, img in (("textured+noise", timg), ("euroc 0_0.jpg", real)):
    a, xy = t(lambda: m.good_features_to_track(img, 0, 0.01, 0.0))
    print(f"{name}: candidates (no min-distance, unlimited) {len(xy)}  {a:.3f} ms")
    a, xy = t(lambda: m.good_features_to_track(img, 1500, 0.01, 8.0)); print(f"  gftt 1500/8: {len(xy)} corners {a:.3f} ms")
    a, xy = t(lambda: m.detect_keypoints(img, 1500)); print(f"  detect_keypoints: {len(xy)} corners {a:.3f} ms")
    b, _ = t(lambda: m.orb_describe(img, xy)); print(f"  orb_describe: {b:.3f} ms")
    c, r = t(lambda: m.detect_and_describe(img, 1500)); print(f"  detect_and_describe: {c:.3f} ms")
