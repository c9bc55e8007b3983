"""On-disk formats: cereal binary / JSON archives of the path's types,
restated in visual-slam_amd/cereal_io.py.  Pins: byte-identical re-serialisation of the
reference's own binary map file and its JSON fixtures (only where /root/reference exists),
the committed golden arrays derived from them, and self round trips."""
import importlib
import json
import os

import numpy as np
import pytest

from tests import cases

cio = importlib.import_module("visual-slam_amd.cereal_io")
synth = importlib.import_module("visual-slam_amd.synth")
REF = "/root/reference/test"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference is only in the build container")


def test_bitset_byte_order():
    d = np.zeros((3, 4), dtype=np.uint64)
    d[0, 0] = 1            # bit 0   -> byte 0, mask 0x80
    d[1, 0] = 1 << 9       # bit 9   -> byte 1, mask 0x40
    d[2, 3] = 1 << 63      # bit 255 -> byte 31, mask 0x01
    raw = cio.unpack_bits_msb_bytes(d)
    assert raw[0, 0] == 0x80 and raw[1, 1] == 0x40 and raw[2, 31] == 0x01 and raw.sum() == 0x80 + 0x40 + 1
    assert np.array_equal(cio.pack_bits_msb_bytes(raw), d)
    s = cio.packed_to_bitstring(d[2])
    assert s[0] == "1" and s.count("1") == 1   # to_string(): char 0 is bit 255
    assert np.array_equal(cio.bitstring_to_packed(s), d[2])
    r = synth.random_descriptors(100, 3)
    assert np.array_equal(cio.pack_bits_msb_bytes(cio.unpack_bits_msb_bytes(r)), r)
    assert all(np.array_equal(cio.bitstring_to_packed(cio.packed_to_bitstring(x)), x) for x in r[:10])


def _toy_map():
    kd = lambda n, s: {"corners": np.arange(2 * n, dtype=np.float64).reshape(n, 2) + 0.25 * s,
                       "corner_angles": np.linspace(-3, 3, n), "corner_descriptors": synth.random_descriptors(n, s)}
    T = np.array([0.1, -0.2, 0.3, 0.0, 0.0, 0.0, 1.0])
    return {
        "corners": [((0, 0), kd(5, 1)), ((0, 1), kd(0, 2)), ((7, 1), kd(3, 3))],
        "matches": [(((0, 0), (0, 1)), {"T_i_j": T, "inliers": np.array([[1, 2]], np.int32),
                                        "matches": np.array([[1, 2], [3, 0]], np.int32)}),
                    (((0, 0), (7, 1)), {"T_i_j": cio.IDENTITY_SE3, "inliers": np.zeros((0, 2), np.int32),
                                        "matches": np.zeros((0, 2), np.int32)})],
        "feature_tracks": [(5, [((0, 0), 1), ((0, 1), 2)]), (-9, [])],
        "outlier_tracks": [],
        "cameras": [((0, 0), cio.IDENTITY_SE3), ((0, 1), T)],
        "landmarks": [(5, {"p": np.array([1.0, 2.0, 3.0]), "obs": [((0, 0), 1)], "outlier_obs": [((0, 1), 2)]})],
    }


def test_binary_round_trip_and_layout(tmp_path):
    m = _toy_map()
    raw = cio.dump_map_binary(m)
    # size tag + first key: 3 images, FrameCamId (0, 0), 5 corners
    assert raw[:8] == (3).to_bytes(8, "little") and raw[8:24] == bytes(16) and raw[24:32] == (5).to_bytes(8, "little")
    p = tmp_path / "map.cereal"
    p.write_bytes(raw)
    back = cio.load_map_binary(str(p))
    assert cio.dump_map_binary(back) == raw
    assert np.array_equal(back["corners"][2][1]["corner_descriptors"], m["corners"][2][1]["corner_descriptors"])
    assert back["feature_tracks"] == m["feature_tracks"]
    (tmp_path / "bad.cereal").write_bytes(raw + b"\0")
    with pytest.raises(ValueError):
        cio.load_map_binary(str(tmp_path / "bad.cereal"))
    (tmp_path / "short.cereal").write_bytes(raw[:-3])
    with pytest.raises(ValueError):
        cio.load_map_binary(str(tmp_path / "short.cereal"))
    cio.save_corners_binary(str(tmp_path / "corners.cereal"), m["corners"])
    cio.save_matches_binary(str(tmp_path / "matches.cereal"), m["matches"])
    c = cio.load_corners_binary(str(tmp_path / "corners.cereal"))
    mm = cio.load_matches_binary(str(tmp_path / "matches.cereal"))
    assert [f for f, _ in c] == [(0, 0), (0, 1), (7, 1)] and np.array_equal(mm[0][1]["matches"], [[1, 2], [3, 0]])
    fcids, desc, off = cio.corners_to_database(c)
    assert list(off) == [0, 5, 5, 8] and desc.shape == (8, 4) and fcids[2] == (7, 1)


def test_json_round_trip():
    m = _toy_map()
    kd = m["corners"][0][1]
    back = cio.keypoints_from_json(json.loads(cio.keypoints_to_json(kd)))
    for k in kd:
        assert np.array_equal(back[k], kd[k])
    md = m["matches"][0][1]
    back = cio.match_data_from_json(json.loads(cio.match_data_to_json(md)))
    assert np.array_equal(back["matches"], md["matches"]) and np.array_equal(back["T_i_j"], md["T_i_j"])


@needs_ref
def test_reference_map_file_reserialises_byte_identically():
    """test/ex4_test_data/map_localize_camera.cereal (loaded by test_ex5.cpp:162): every byte
    is consumed and writing the parsed map back reproduces the file."""
    p = os.path.join(REF, "ex4_test_data/map_localize_camera.cereal")
    m = cio.load_map_binary(p)
    assert [len(m[k]) for k in cio.MAP_SECTIONS] == [20, 190, 811, 20, 7, 229]
    with open(p, "rb") as f:
        assert cio.dump_map_binary(m) == f.read()


@needs_ref
def test_reference_json_fixtures_equal_committed_goldens():
    d1, d2 = cases.ex3()
    kd0 = cio.load_keypoints_json(os.path.join(REF, "ex3_test_data/kd0.json"))
    kd1 = cio.load_keypoints_json(os.path.join(REF, "ex3_test_data/kd1.json"))
    assert np.array_equal(kd0["corner_descriptors"], d1) and np.array_equal(kd1["corner_descriptors"], d2)
    assert kd0["corners"].shape == (537, 2) and kd0["corner_angles"].shape == (537,)
    md = cio.load_match_data_json(os.path.join(REF, "ex3_test_data/matches.json"))
    assert np.array_equal(md["matches"], np.load(os.path.join(cases.GOLDEN, "ex3_matches.npy")))
    # our JSON writer reproduces the reference's file value for value
    with open(os.path.join(REF, "ex3_test_data/kd0.json")) as f:
        assert json.loads(cio.keypoints_to_json(kd0)) == json.load(f)


def test_cpp_cereal_lite_agrees_with_python(tmp_path):
    """include/visnav_amd/cereal_lite.h (C++, no cereal): reads a corners.cereal written here,
    writes it back byte-identically, and its matches.cereal parses to what the rule says."""
    import subprocess
    exe = os.path.join(os.path.dirname(cases.GOLDEN.rstrip("/")), "cpp", "cereal_lite_test")
    if not os.path.exists(exe):
        pytest.skip("run __graft_entry__.build() first")
    m = _toy_map()
    src, back, mout = (str(tmp_path / n) for n in ("corners.cereal", "corners_back.cereal", "matches.cereal"))
    cio.save_corners_binary(src, m["corners"])
    r = subprocess.run([exe, src, back, mout], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(src, "rb").read() == open(back, "rb").read()
    got = cio.load_matches_binary(mout)
    assert len(got) == len(m["corners"]) - 1
    for i, ((f1, f2), md) in enumerate(got, start=1):
        assert f1 == m["corners"][i][0] and f2 == m["corners"][i - 1][0]
        d = m["corners"][i - 1][1]["corner_descriptors"]
        want = [[k, int(sum(bin(int(w)).count("1") for w in d[k]))] for k in range(len(d))]
        assert md["matches"].tolist() == want and md["inliers"].tolist() == want[:1]
        assert np.array_equal(md["T_i_j"], cio.IDENTITY_SE3)
    if os.path.isdir(REF):  # the corners section of the reference's own map file
        ref = cio.load_map_binary(os.path.join(REF, "ex4_test_data/map_localize_camera.cereal"))["corners"]
        cio.save_corners_binary(src, ref)
        r = subprocess.run([exe, src, back, mout], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and open(src, "rb").read() == open(back, "rb").read()
