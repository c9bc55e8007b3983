"""BASELINE config 3 on real EuRoC frames (the subset that travels: tests/golden/euroc_subset/,
8 images of the reference's data/euroc_V1; goldens = the reference's own feature_corners from
its binary map file + its real matchDescriptors, see tests/golden/make_golden_euroc.py).
CPU: the C port reproduces the goldens.  GPU: the whole pipeline -- detection, orientation,
descriptors, stereo and all-pairs matching -- reproduces them bit for bit."""
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

from oracle import binding
from tests import cases

DIR = os.path.join(cases.GOLDEN, "euroc_subset")
with open(os.path.join(DIR, "golden.json")) as f:
    G = json.load(f)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def image(k):
    from PIL import Image
    return np.array(Image.open(os.path.join(DIR, G["images"][k]["file"])))


def test_port_reproduces_goldens():
    o = binding.port()
    kds = []
    for k, rec in enumerate(G["images"]):
        xy = o.detect_keypoints(image(k), 1500)
        _, desc = o.orb_describe(image(k), xy, True)
        assert len(xy) == rec["corners"] and sha(xy) == rec["corners_sha256"] and sha(desc) == rec["descriptors_sha256"]
        kds.append(desc)
    for p in G["pairs"]:
        m = o.match(kds[p["i"]], kds[p["j"]], 70, 1.2)
        assert len(m) == p["matches"] and sha(m.astype(np.int32)) == p["sha256"]


@pytest.mark.gpu
def test_gpu_pipeline_reproduces_goldens():
    vsm = importlib.import_module("visual-slam_amd")
    with vsm.Matcher(0) as m:
        kds = []
        for k, rec in enumerate(G["images"]):
            xy, ang, desc = m.detect_and_describe(image(k), 1500, True)   # sfm.cpp:1199
            assert len(xy) == rec["corners"] and sha(xy) == rec["corners_sha256"]
            assert sha(desc) == rec["descriptors_sha256"]
            kds.append(desc)
        # match_stereo (sfm.cpp:1234-1244): one call per stereo pair
        for p in (p for p in G["pairs"] if p["kind"] == "stereo"):
            got = m.match(kds[p["i"]], kds[p["j"]], 70, 1.2)
            assert len(got) == p["matches"] and sha(got) == p["sha256"]
        # match_all (sfm.cpp:1276-1321): the whole pair list as one batch
        off = np.zeros(len(kds) + 1, np.uint32)
        off[1:] = np.cumsum([len(d) for d in kds])
        m.db_upload(np.concatenate(kds), off)
        allp = [p for p in G["pairs"] if p["kind"] == "all"]
        offs, lists = m.match_pairs(np.array([[p["i"], p["j"]] for p in allp], np.uint32), 70, 1.2)
        for k, p in enumerate(allp):
            got = lists[int(offs[k]): int(offs[k + 1])]
            assert len(got) == p["matches"] and sha(got) == p["sha256"], p


@pytest.mark.gpu
def test_cpp_tools_images_in_matches_file_out(tmp_path):
    """The same pipeline from C++, file to file: examples/vsm_sfm_detect (the 8 frames as PGM files ->
    corners.cereal, detect_keypoints() of sfm.cpp:1191-1214 in one batch call) and examples/vsm_sfm_match
    (-> matches.cereal, match_stereo / match_all): the files hold the goldens -- the reference's own corner
    lists and descriptors and its real matchDescriptors' lists."""
    import subprocess
    cio = importlib.import_module("visual-slam_amd.cereal_io")
    ex = os.path.join(os.path.dirname(cases.GOLDEN), "..", "examples")
    detect, match = os.path.join(ex, "vsm_sfm_detect"), os.path.join(ex, "vsm_sfm_match")
    assert os.path.exists(detect) and os.path.exists(match), "run __graft_entry__.build() first"
    lines = []
    for k in range(len(G["images"])):
        im = image(k)
        p = tmp_path / f"{k}.pgm"
        with open(p, "wb") as f:
            f.write(b"P5\n# frame %d\n%d %d\n255\n" % (k, im.shape[1], im.shape[0]) + np.ascontiguousarray(im).tobytes())
        lines.append(f"{k // 2} {k % 2} {p}")
    (tmp_path / "images.txt").write_text("\n".join(lines) + "\n")
    corners = str(tmp_path / "corners.cereal")
    r = subprocess.run([detect, str(tmp_path / "images.txt"), corners, "1500", "1"], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0 and "vsm_sfm_detect: 8 images" in r.stdout, r.stdout + r.stderr
    items = cio.load_corners_binary(corners)
    assert [fc for fc, _ in items] == [(k // 2, k % 2) for k in range(8)]
    for (fc, kd), rec in zip(items, G["images"]):
        assert len(kd["corners"]) == rec["corners"] and sha(np.asarray(kd["corners"], np.float64)) == rec["corners_sha256"]
        assert sha(np.asarray(kd["corner_descriptors"], np.uint64)) == rec["descriptors_sha256"]
    for mode, kind in (("stereo", "stereo"), ("all", "all")):
        out = str(tmp_path / (mode + ".cereal"))
        r = subprocess.run([match, corners, out, mode], capture_output=True, text=True, timeout=100)
        assert r.returncode == 0, r.stdout + r.stderr
        back = {(a, b): md for (a, b), md in cio.load_matches_binary(out)}
        want = [p for p in G["pairs"] if p["kind"] == kind]
        assert len(back) == len(want)
        for p in want:
            md = back[((p["i"] // 2, p["i"] % 2), (p["j"] // 2, p["j"] % 2))]
            got = np.asarray(md["matches"], np.int32).reshape(-1, 2)
            assert len(got) == p["matches"] and sha(got) == p["sha256"], p
