"""GPU parity of vsm_detect_keypoints / vsm_good_features_to_track / vsm_detect_and_describe
(detectKeypoints and detectKeypointsAndDescriptors, keypoints.h:133-149, 239-245) through the C
ABI: the reference's golden corner lists and the CPU oracle.  Bar: identical corner lists
(positions and order), descriptors bit-exact, angles as in tests/test_gpu_orb.py."""
import importlib
import os

import numpy as np
import pytest

from oracle import binding
from tests import cases
from tests.test_gftt_oracle import GFTT_CASES
from tests.test_orb_oracle import ex3_image, ex3_kd, synth_image

pytestmark = pytest.mark.gpu
vsm = importlib.import_module("visual-slam_amd")


@pytest.fixture(scope="module")
def m():
    with vsm.Matcher(0) as mm:
        yield mm


@pytest.mark.parametrize("k", [0, 1])
def test_reference_corner_lists(m, k):
    want = ex3_kd(k)[0]
    got = m.detect_keypoints(ex3_image(k), 1500)
    assert got.shape == want.shape and np.array_equal(got, want)
    few = m.detect_keypoints(ex3_image(k), 100)
    assert np.array_equal(few, binding.port().detect_keypoints(ex3_image(k), 100))


def test_keypoints_all_on_the_gpu(m):
    """The reference's KeypointsAll test (test_ex3.cpp:187-218): both images through
    detectKeypointsAndDescriptors and matchDescriptors(70, 1.2) -- every stage on the GPU --
    give the golden kd0.json / kd1.json / matches.json."""
    d = []
    for k in (0, 1):
        xy, ang, desc = m.detect_and_describe(ex3_image(k), 1500, True)
        wxy, wang, wdesc = ex3_kd(k)
        assert np.array_equal(xy, wxy) and np.abs(ang - wang).max() < 1e-8 and np.array_equal(desc, wdesc)
        d.append(desc)
    want = np.load(os.path.join(cases.GOLDEN, "ex3_matches.npy"))
    assert np.array_equal(m.match(d[0], d[1], 70, 1.2), want)


@pytest.mark.parametrize("case", GFTT_CASES + [(752, 480, 21, 1500, 0.01, 8.0), (1920, 1080, 22, 5000, 0.01, 8.0),
                                               (1000, 31, 23, 0, 0.02, 1.0), (640, 480, 24, 0, 0.01, 0.0)])
def test_good_features_equal_oracle(m, case):
    w, h, seed, mc, q, md = case
    img = synth_image(w, h, seed) if min(w, h) >= 8 else np.random.default_rng(seed).integers(0, 256, (h, w)).astype(np.uint8)
    want = binding.port().gftt(img, mc, q, md)
    got = m.good_features_to_track(img, mc, q, md)
    assert got.shape == want.shape and np.array_equal(got, want), (case, len(got), len(want))


def test_noise_flat_and_tie_images(m):
    o = binding.port()
    noise = np.random.default_rng(5).integers(0, 256, (300, 400)).astype(np.uint8)  # ~1/9 of the pixels are local maxima
    assert np.array_equal(m.good_features_to_track(noise, 0, 0.01, 2.0), o.gftt(noise, 0, 0.01, 2.0))
    assert np.array_equal(m.detect_keypoints(noise, 1500), o.detect_keypoints(noise, 1500))
    assert len(m.good_features_to_track(np.full((40, 50), 77, np.uint8), 100)) == 0
    yy, xx = np.indices((48, 64))
    chk = ((yy // 8 + xx // 8) % 2 * 255).astype(np.uint8)  # exactly equal eigenvalues: the tie order matters
    got = m.good_features_to_track(chk, 0, 0.01, 4.0)
    assert len(got) > 10 and np.array_equal(got, o.gftt(chk, 0, 0.01, 4.0))
    xy, ang, desc = m.detect_and_describe(noise[:, 3:395], 800, True)       # strided rows
    oxy = o.detect_keypoints(np.ascontiguousarray(noise[:, 3:395]), 800)
    oa, od = o.orb_describe(np.ascontiguousarray(noise[:, 3:395]), oxy)
    assert np.array_equal(xy, oxy) and np.array_equal(ang, oa) and np.array_equal(desc, od)


def test_errors_and_capacity(m):
    import ctypes as C
    img = synth_image(96, 64, 1)
    with pytest.raises(vsm.VsmError):
        m.good_features_to_track(img, 10, 0.0, 8.0)      # CV_Assert(qualityLevel > 0)
    with pytest.raises(vsm.VsmError):
        m.good_features_to_track(img, 10, 0.01, -1.0)
    n = C.c_size_t(0)
    out = np.zeros((2, 2))
    rc = vsm.matcher.lib().vsm_detect_keypoints(m._h, C.c_void_p(img.ctypes.data), 96, 64, 96, 0, C.c_void_p(out.ctypes.data),
                                                2, C.byref(n))
    assert rc == vsm.matcher.VSM_ERR_CAPACITY and n.value == len(binding.port().detect_keypoints(img, 0)) > 2


def test_cpp_shim_detect_and_describe(tmp_path):
    """detectKeypointsAndDescriptors with the reference's argument list (keypoints.h:239-241)
    over stand-in Pangolin/KeypointsData types; binary built by build()."""
    import subprocess
    exe = os.path.join(os.path.dirname(cases.GOLDEN.rstrip("/")), "cpp", "shim_orb_test")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    xy, angles, desc = ex3_kd(1)
    img = ex3_image(1)
    np.ascontiguousarray(img).tofile(str(tmp_path / "img.bin"))
    np.array([img.shape[1], img.shape[0]], np.int32).tofile(str(tmp_path / "dims.bin"))
    xy.tofile(str(tmp_path / "corners.bin"))
    angles.tofile(str(tmp_path / "angles.bin"))
    desc.tofile(str(tmp_path / "desc.bin"))
    r = subprocess.run([exe, str(tmp_path), "detect"], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr


def test_repeated_detection_is_bit_identical(m):
    """Race check: candidate compaction uses an atomic counter (arbitrary order) and the device
    rank sort must undo it -- 20 runs, identical corners, angles and descriptors."""
    img = np.random.default_rng(11).integers(0, 256, (480, 752)).astype(np.uint8)
    ref = m.detect_and_describe(img, 1500, True)
    for _ in range(20):
        got = m.detect_and_describe(img, 1500, True)
        assert all(np.array_equal(a, b) for a, b in zip(got, ref))


def _batch_images():
    rng = np.random.default_rng(17)
    imgs = [synth_image(752, 480, 40 + k) for k in range(5)]
    imgs.append(rng.integers(0, 256, (480, 752)).astype(np.uint8))     # noise: > 4096 candidates wanted by the pass
    imgs.append(np.full((480, 752), 90, np.uint8))                      # flat: no corner at all
    imgs.append(ex3_image(0) if ex3_image(0).shape == (480, 752) else synth_image(752, 480, 3))
    return imgs


@pytest.mark.parametrize("num_features,rotate", [(1500, True), (200, False), (0, True)])
def test_batch_front_end_equals_image_by_image(m, num_features, rotate):
    """vsm_detect_and_describe_batch (the loop of detect_keypoints(), sfm.cpp:1191-1201, in one
    call) against vsm_detect_and_describe image by image: offsets, corners, angles and descriptors
    bit for bit; one image against the oracle as well."""
    imgs = _batch_images()
    off, xy, ang, desc = m.detect_and_describe_batch(imgs, num_features, rotate)
    assert off[0] == 0 and off[-1] == len(xy) == len(ang) == len(desc)
    for k, im in enumerate(imgs):
        wxy, wang, wdesc = m.detect_and_describe(im, num_features, rotate)
        s = slice(int(off[k]), int(off[k + 1]))
        assert np.array_equal(xy[s], wxy) and np.array_equal(ang[s], wang) and np.array_equal(desc[s], wdesc), k
    assert off[7] - off[6] == 0 and off[1] - off[0] > 100
    o = binding.port()
    oxy = o.detect_keypoints(imgs[2], num_features)
    oa, od = o.orb_describe(imgs[2], oxy, rotate)
    s = slice(int(off[2]), int(off[3]))
    assert np.array_equal(xy[s], oxy) and np.array_equal(ang[s], oa) and np.array_equal(desc[s], od)


def test_batch_front_end_into_the_database(m):
    """Images in, match lists out: the batch front end leaves the descriptors on the device as the
    context's database (keep_as_database), vsm_match_pairs follows with no upload -- equal to the
    oracle's matchDescriptors on the descriptors the batch also returned; strided rows; a second
    call replaces the database; only the wanted arrays are written."""
    imgs = [synth_image(400, 300, 60 + k)[:, 5:395] for k in range(6)]   # pitch 400, width 390
    off, xy, ang, desc = m.detect_and_describe_batch(imgs, 600, True, keep_as_database=True)
    assert off[-1] > 600
    pairs = cases.synth.all_pairs(len(imgs))
    offs, lists = m.match_pairs(pairs, 90, 1.1)
    o = binding.port()
    for k, (i, j) in enumerate(pairs.tolist()):
        want = o.match(desc[off[i]:off[i + 1]], desc[off[j]:off[j + 1]], 90, 1.1)
        assert np.array_equal(lists[int(offs[k]):int(offs[k + 1])], want), (i, j)
    off2, xy2, ang2, desc2 = m.detect_and_describe_batch(imgs[:2], 300, False, keep_as_database=True, want=("descriptors",))
    assert xy2 is None and ang2 is None and len(desc2) == off2[-1]
    offs, lists = m.match_pairs(np.array([[0, 1]], np.uint32), 90, 1.1)
    assert np.array_equal(lists, o.match(desc2[:off2[1]], desc2[off2[1]:], 90, 1.1))


def test_batch_front_end_in_several_chunks(m):
    """More images than one chunk of device work areas holds (forced by VSM_BATCH_CHUNK: 3 per chunk
    instead of ~1000): same results, same kept database."""
    imgs = _batch_images()
    want = m.detect_and_describe_batch(imgs, 700, True)
    os.environ["VSM_BATCH_CHUNK"] = "3"
    try:
        got = m.detect_and_describe_batch(imgs, 700, True, keep_as_database=True)
    finally:
        del os.environ["VSM_BATCH_CHUNK"]
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    off, desc = got[0], got[3]
    pairs = np.array([[0, 1], [4, 2], [7, 5], [3, 6], [5, 7]], np.uint32)
    offs, lists = m.match_pairs(pairs, 80, 1.2)
    for k, (i, j) in enumerate(pairs.tolist()):
        assert np.array_equal(lists[int(offs[k]):int(offs[k + 1])],
                              binding.port().match(desc[off[i]:off[i + 1]], desc[off[j]:off[j + 1]], 80, 1.2))


def test_batch_front_end_errors_and_capacity(m):
    import ctypes as C
    L = vsm.matcher.lib()
    imgs = [synth_image(96, 64, 1), synth_image(96, 64, 2)]
    ptrs = (C.c_void_p * 2)(*[im.ctypes.data for im in imgs])
    off = np.zeros(3, np.uint32)
    xy = np.zeros((4, 2))
    rc = L.vsm_detect_and_describe_batch(m._h, ptrs, 2, 96, 64, 96, 50, 1, C.c_void_p(off.ctypes.data), C.c_void_p(xy.ctypes.data),
                                         None, None, 4, 0)
    want = [len(binding.port().detect_keypoints(im, 50)) for im in imgs]
    assert rc == vsm.matcher.VSM_ERR_CAPACITY and list(np.diff(off)) == want and sum(want) > 4
    assert L.vsm_detect_and_describe_batch(m._h, ptrs, 2, 96, 64, 96, 0, 1, C.c_void_p(off.ctypes.data), None, None, None, 0,
                                           1) == vsm.matcher.VSM_ERR_INVALID      # a kept database needs num_features > 0
    assert L.vsm_detect_and_describe_batch(m._h, ptrs, 2, 96, 64, 95, 50, 1, C.c_void_p(off.ctypes.data), None, None, None, 0,
                                           0) == vsm.matcher.VSM_ERR_INVALID      # pitch < width
    assert L.vsm_detect_and_describe_batch(m._h, ptrs, 2, 96, 64, 96, 50, 1, None, None, None, None, 0, 0) == vsm.matcher.VSM_ERR_INVALID
    # too small with keep_as_database: no database is kept (and the one before is gone, as after a failed upload)
    m.db_upload(np.zeros((4, 4), np.uint64), np.array([0, 2, 4], np.uint32))
    rc = L.vsm_detect_and_describe_batch(m._h, ptrs, 2, 96, 64, 96, 50, 1, C.c_void_p(off.ctypes.data), C.c_void_p(xy.ctypes.data),
                                         None, None, 4, 1)
    assert rc == vsm.matcher.VSM_ERR_CAPACITY
    with pytest.raises(vsm.VsmError):
        m.match_pairs(np.array([[0, 1]], np.uint32))
    # counts only (no array wanted), and the empty batch
    assert L.vsm_detect_and_describe_batch(m._h, ptrs, 2, 96, 64, 96, 50, 1, C.c_void_p(off.ctypes.data), None, None, None, 0, 0) == 0
    assert list(np.diff(off)) == want
    off0, xy0, ang0, desc0 = m.detect_and_describe_batch([], 100)
    assert list(off0) == [0] and len(xy0) == 0
