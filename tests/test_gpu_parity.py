"""GPU parity: the HIP path through the C ABI against the oracle, the recorded
outputs of the real reference function and the reference's own golden vector.
Bar: bit-exact (integer/index work)."""
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

from oracle import binding
from tests import cases

pytestmark = pytest.mark.gpu
vsm = importlib.import_module("visual-slam_amd")

with open(os.path.join(cases.GOLDEN, "ref_cases.json")) as f:
    REF = json.load(f)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int32).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def m():
    with vsm.Matcher(0) as mm:
        yield mm


def test_ex3_golden_vector(m):
    """test_ex3.cpp:148-185 on the GPU, order included."""
    d1, d2 = cases.ex3()
    want = np.load(os.path.join(cases.GOLDEN, "ex3_matches.npy"))
    assert np.array_equal(m.match(d1, d2, 70, 1.2), want)
    assert np.array_equal(vsm.match_descriptors(d1, d2, 70, 1.2), want)


@pytest.mark.parametrize("name", cases.case_names())
def test_pair_matches_recorded_reference(m, name):
    d1, d2, thr, ratio = cases.build_case(name)
    rec = REF["cases"][name]
    got = m.match(d1, d2, thr, ratio)
    assert len(got) == rec["count"]
    assert sha(got) == rec["sha256"]
    assert np.array_equal(got, binding.port().match(d1, d2, thr, ratio))


@pytest.mark.parametrize("variant", [1, 2, 4, 8, 0x20001, 0x20002, 0x20004, 0x20008, 0x0301, 0x20304, 0x4008,
                                     0x40008, 0x60008, 0x40308, 0x80001, 0x80004, 0x80008, 0xC0008, 0x1000008, 0x1020008,
                                     0x0308, 0x2000008, 0x2000308, 0x10000008, 0x20000008, 0x40000008, 0x12000008, 0x20000208, 0x4000008, 0x8000001, 0x8000000,
                                     0xE00008, 0xD00008, 0xC00008, 0x10E00008, 0x20E00008, 0xE00308, 0xE00001, 0xD00001])
def test_pair_all_kernel_shapes(m, variant):
    """Every instantiation (queries per lane 1/2/4/8 x float-key / integer-key
    tracking [bit 17], with forced train splits [bits 8-15]; R = 8 runs the adaptive tile
    decomposition unless bit 18 is set -- on the matrix cores (mfma_scan.h) unless bit 25 asks
    for the popcount scan; bits 28-30 force the tile size; bits 19 / 24 select earlier popcount
    forms: 512-query tiles without the skip test, one-wave / 4-wave workgroups; nibble 20-23 =
    14 / 13 / 12 forces the distance form / the key form / distances in forward scans only) gives
    the same list."""
    m.set_variant(variant)
    try:
        for name in ("ex3_thr70_r1.2", "planted_300x777", "planted_1023x1025", "planted_4100x130",
                     "edge_all_identical", "edge_all_256_thr300_ratio1", "edge_tie_best"):
            d1, d2, thr, ratio = cases.build_case(name)
            assert sha(m.match(d1, d2, thr, ratio)) == REF["cases"][name]["sha256"], (variant, name)
    finally:
        m.set_variant(0)


@pytest.mark.parametrize("thr,ratio", [(70, 1.2), (256, 1.0), (300, 1.0), (257, 0.9), (50, 3.0), (100, 0.5), (1, 1.2),
                                       (0, 1.2), (-5, 1.2), (70, 0.0), (70, -1.0), (70, 4.0), (200, 1.01), (70, float("nan")),
                                       (128, 1.5), (255, 1.0)])
def test_accept_limit_of_the_matrix_core_scan(m, thr, ratio):
    """The matrix-core scan starts (best, second) at the smallest distance the accept test
    cannot tell from anything larger (accept_limit, vsmatch.hip); whatever threshold and ratio
    that limit is derived from, the list is the oracle's."""
    for variant in (0xD00008, 0x10D00008, 0xD00308, 0xE00008, 0x10E00008, 0xE00308):
        m.set_variant(variant)
        try:
            for (n1, n2, seed) in [(300, 777, 1), (1023, 1025, 2), (64, 33, 3), (5, 2000, 4)]:
                d1, d2 = cases.synth.orb_like_pair(n1, n2, seed)
                assert np.array_equal(m.match(d1, d2, thr, ratio), binding.port().match(d1, d2, thr, ratio)), \
                    (variant, thr, ratio, n1, n2)
        finally:
            m.set_variant(0)


def _real_like_database(n_img, n_desc, seed):
    """Images drawn from the reference's 1084 real ORB descriptors (ex3 kd0 + kd1), a few bits
    flipped per copy: unrelated descriptors are 126 +- 19 bits apart, as in real images."""
    real = np.concatenate(cases.ex3())
    rng = np.random.default_rng(seed)
    parts = [real[rng.choice(len(real), size=n_desc, replace=n_desc > len(real))]
             ^ cases.synth.flip_masks(n_desc, seed, 700 + i, levels=(5,)) for i in range(n_img)]
    db = np.ascontiguousarray(np.concatenate(parts))
    return db, (np.arange(n_img + 1) * n_desc).astype(np.uint32)


@pytest.mark.parametrize("variant", [0xE00008, 0x10E00008, 0x20E00008, 0xE00308, 0xD00008, 0xE00001, 0xD00001])
def test_real_descriptors_every_tail_length(m, variant):
    """Real ORB descriptors (many near neighbours: the slow paths run in nearly every step) at
    train sizes around the 32-row step of the matrix-core scan -- the partial last step is handled
    apart from the pipelined loop -- under both tracking forms; also raw 2-NN."""
    real = np.concatenate(cases.ex3())
    noisy = real ^ cases.synth.flip_masks(len(real), 3, 77, levels=(5, 6))
    m.set_variant(variant)
    try:
        for n1, n2 in [(200, 1), (200, 31), (200, 32), (200, 33), (150, 63), (150, 64), (150, 65), (97, 95), (64, 96),
                       (33, 127), (1084, 1000), (1000, 1084), (5, 1084)]:
            d1, d2 = real[:n1], noisy[len(noisy) - n2:]
            assert np.array_equal(m.match(d1, d2, 70, 1.2), binding.port().match(d1, d2, 70, 1.2)), (hex(variant), n1, n2)
            for g, w in zip(m.nn2(d1, d2), binding.port().nn2(d1, d2)):
                assert np.array_equal(g, w), (hex(variant), n1, n2)
    finally:
        m.set_variant(0)


def test_tracking_form_follows_the_data_and_never_changes_the_lists():
    """The matrix-core forward scan has two tracking forms (mfma_scan.h); the context picks one
    after every batched call from the share of steps that left the common path.  Real-like
    descriptors -> distances, iid random ones -> keys; the lists equal the oracle's under either
    form, forced or chosen, pruned backward pass included."""
    with vsm.Matcher(0) as mm:
        assert mm.scan_tracking() == (1, -1.0)
        db, off = _real_like_database(64, 900, 5)  # 2016 pairs: enough waves for the batched (adaptive) scan
        pairs = cases.synth.all_pairs(64)
        mm.db_upload(db, off)
        want = [binding.port().match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2) for i, j in pairs.tolist()]
        assert sum(len(w) for w in want) > 10000
        for variant in (0, 0, 0xE00000, 0xD00000, 0xC00000, 0x2000000):
            mm.set_variant(variant)
            offs, out = mm.match_pairs(pairs, 70, 1.2)
            for k, w in enumerate(want):
                assert np.array_equal(out[int(offs[k]):int(offs[k + 1])], w), (hex(variant), k)
            mode, share = mm.scan_tracking()
            if variant != 0x2000000:
                assert mode == 1 and share > 0.5, (hex(variant), mode, share)
        mm.set_variant(0)
        db, off = cases.synth.sfm_database(70, 1000, seed=12)
        pairs = cases.synth.all_pairs(70)
        mm.db_upload(db, off)
        o1, l1 = mm.match_pairs(pairs, 70, 1.2)      # still the distance form
        mode, share = mm.scan_tracking()
        assert mode == 0 and 0 <= share < 0.05, (mode, share)
        o2, l2 = mm.match_pairs(pairs, 70, 1.2)      # now keys
        assert np.array_equal(o1, o2) and np.array_equal(l1, l2)
        for k in range(0, len(pairs), 37):
            i, j = pairs[k]
            assert np.array_equal(l2[int(o2[k]):int(o2[k + 1])],
                                  binding.port().match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2))


def test_nn2_vs_oracle(m):
    for variant in (0, 0x20000, 8, 0x2000008, 0x10000008, 0xD00008, 0x10D00008, 0xE00001):
        m.set_variant(variant)
        try:
            for (nq, nt, seed) in [(1, 1, 1), (5, 9, 2), (64, 64, 3), (257, 1000, 4), (1000, 257, 5), (777, 3001, 6),
                                   (300, 40000, 7), (130, 70001, 8)]:  # the last two exceed one float-key range
                q, t = cases.synth.orb_like_pair(nq, nt, seed)
                got = m.nn2(q, t)
                want = binding.port().nn2(q, t)
                for g, w in zip(got, want):
                    assert np.array_equal(g, w), (variant, nq, nt)
        finally:
            m.set_variant(0)
    q, _ = cases.synth.orb_like_pair(10, 10, 9)
    bi, bd, sd = m.nn2(q, np.zeros((0, 4), np.uint64))
    assert list(bi) == [-1] * 10 and list(bd) == [256] * 10 and list(sd) == [256] * 10
    bi, bd, sd = m.nn2(q[:1], q[:1] ^ cases.ONES)
    assert (bi[0], bd[0], sd[0]) == (-1, 256, 256)


def test_batched_small_ragged_database(m):
    """match_all pattern (sfm.cpp:1276-1321) on a ragged database incl. empty images."""
    db, off, pairs = cases.sfm_small()
    m.db_upload(db, off)
    offs, out = m.match_pairs(pairs, 70, 1.2)
    recs = REF["sfm_small"]["pairs"]
    assert int(offs[0]) == 0 and int(offs[-1]) == len(out)
    for k, r in enumerate(recs):
        got = out[int(offs[k]):int(offs[k + 1])]
        assert len(got) == r["count"] and sha(got) == r["sha256"], (k, r["pair"])
    # both orientations and repeated pairs
    pp = np.concatenate([pairs[:, ::-1], pairs[:3]])
    offs, out = m.match_pairs(pp, 70, 1.2)
    for k, (i, j) in enumerate(pp.tolist()):
        want = binding.port().match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2)
        assert np.array_equal(out[int(offs[k]):int(offs[k + 1])], want)
    offs, out = m.match_pairs(np.zeros((0, 2), np.uint32))
    assert list(offs) == [0] and len(out) == 0


def test_batched_uniform_database_vs_oracle(m):
    db, off = cases.synth.sfm_database(24, 700, seed=33)
    pairs = cases.synth.all_pairs(24)
    m.db_upload(db, off)
    offs, out = m.match_pairs(pairs, 70, 1.2)
    total = 0
    for k, (i, j) in enumerate(pairs.tolist()):
        want = binding.port().match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2)
        assert np.array_equal(out[int(offs[k]):int(offs[k + 1])], want), (i, j)
        total += len(want)
    assert total == len(out) and total > 1000  # the planted overlap is really there


def test_pruned_backward_pass_equals_full(m):
    """The backward scan restricted to nominated columns (nominate_kernel) gives the
    same lists as scanning every column; dense overlap, no overlap and ragged sizes."""
    for (n_img, sizes, share, win) in [(12, 900, 0.9, 12), (12, 900, 0.0, 1), (9, [0, 5, 64, 65, 1000, 1, 257, 2000, 333], 0.6, 9)]:
        db, off = cases.synth.sfm_database(n_img, sizes, seed=91, share_frac=share, window=win)
        pairs = cases.synth.all_pairs(n_img)
        pairs = np.concatenate([pairs, pairs[:, ::-1]])
        m.db_upload(db, off)
        m.set_variant(0)
        o1, l1 = m.match_pairs(pairs, 70, 1.2)
        ev_pruned = m.last_scan_stats()["comparisons"]
        m.set_variant(0x10000)
        try:
            o2, l2 = m.match_pairs(pairs, 70, 1.2)
            ev_full = m.last_scan_stats()["comparisons"]
        finally:
            m.set_variant(0)
        assert np.array_equal(o1, o2) and np.array_equal(l1, l2)
        sz = np.diff(off.astype(np.int64))
        assert ev_full == int((2 * sz[pairs[:, 0]] * sz[pairs[:, 1]]).sum())
        assert ev_full // 2 <= ev_pruned <= ev_full
        if share == 0.0:
            assert len(l1) == 0 or ev_pruned < 0.6 * ev_full
        k = 7
        i, j = pairs[k]
        want = binding.port().match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2)
        assert np.array_equal(l1[int(o1[k]):int(o1[k + 1])], want)


def test_capacity_and_errors(m):
    d1, d2, thr, ratio = cases.build_case("planted_300x777")
    L = vsm.matcher.lib()
    import ctypes as C
    out = np.empty((4, 2), np.int32)
    n = C.c_size_t(0)
    rc = L.vsm_match_pair(m._h, d1.ctypes.data, len(d1), d2.ctypes.data, len(d2), thr, ratio,
                          out.ctypes.data, 4, C.byref(n))
    assert rc == vsm.matcher.VSM_ERR_CAPACITY and n.value == REF["cases"]["planted_300x777"]["count"]
    db, off, pairs = cases.sfm_small()
    m.db_upload(db, off)
    with pytest.raises(vsm.VsmError):
        m.match_pairs(np.array([[0, 99]], np.uint32), cap=10)


def test_property_swap_transposes(m):
    """Swapping the arguments gives the transposed set at full size."""
    a, b = cases.synth.orb_like_pair(2000, 1900, seed=77)
    ab = m.match(a, b, 70, 1.2)
    ba = m.match(b, a, 70, 1.2)
    assert len(ab) > 500
    assert sorted(map(tuple, ab.tolist())) == sorted((y, x) for x, y in ba.tolist())
    assert np.all(np.diff(ab[:, 0]) > 0) and np.all(np.diff(ba[:, 0]) > 0)


def test_cpp_shim_reference_signature():
    """visnav::matchDescriptors with the reference's C++ signature (keypoints.h:247-250),
    std::bitset in / std::pair out, concurrent callers; binary built by build()."""
    import subprocess
    root = os.path.dirname(cases.GOLDEN.rstrip("/"))
    exe = os.path.join(root, "cpp", "shim_test")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    g = cases.GOLDEN
    r = subprocess.run([exe, os.path.join(g, "ex3_kd0.npy"), os.path.join(g, "ex3_kd1.npy"),
                        os.path.join(g, "ex3_matches.npy")], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr


def test_config3_shape_euroc_like_all_pairs(m):
    """BASELINE config 3 shape: 164 images (82 stereo frames) of ~1200-1500 descriptors,
    the 13284 non-stereo pairs of sfm.cpp:1285-1290.  Sampled pairs against the oracle;
    all pairs through the argument-swap property (size independent)."""
    rng = np.random.default_rng(3)
    sizes = rng.integers(1200, 1501, size=164)
    db, off = cases.synth.sfm_database(164, sizes, seed=55, share_frac=0.5, window=12)
    frame = np.repeat(np.arange(82), 2)
    pairs = cases.synth.all_pairs(164, frame)
    assert len(pairs) == 13284
    m.db_upload(db, off)
    o1, l1 = m.match_pairs(pairs, 70, 1.2)
    o2, l2 = m.match_pairs(pairs[:, ::-1], 70, 1.2)
    assert int(o1[-1]) == len(l1) == len(l2) and len(l1) > 10000
    assert np.array_equal(np.diff(o1.astype(np.int64)), np.diff(o2.astype(np.int64)))
    # transposed sets pair by pair: sort the swapped list of each pair by its second column
    pid1 = np.repeat(np.arange(len(pairs)), np.diff(o1.astype(np.int64)))
    pid2 = np.repeat(np.arange(len(pairs)), np.diff(o2.astype(np.int64)))
    a = np.stack([pid1, l1[:, 0], l1[:, 1]], axis=1)
    b = np.stack([pid2, l2[:, 1], l2[:, 0]], axis=1)
    b = b[np.lexsort((b[:, 1], b[:, 0]))]
    assert np.array_equal(a, b)
    for k in rng.choice(len(pairs), size=120, replace=False):
        i, j = pairs[k]
        want = binding.port().match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2)
        assert np.array_equal(l1[int(o1[k]):int(o1[k + 1])], want), (k, i, j)


def test_nn2_large_sampled_against_oracle(m):
    """One-direction scan at 200k x 200k (several float-key ranges, R = 8): invariants on
    every query, exact values on a sample."""
    n = 200_000
    q, t = cases.synth.planted_sets(n, seed=8)
    bi, bd, sd = m.nn2(q, t)
    assert bi.min() >= 0 and bi.max() < n
    assert np.all(bd <= sd) and bd.min() >= 0 and sd.max() <= 256
    assert (bd < 70).sum() > n // 3  # the planted half is found
    rng = np.random.default_rng(4)
    pick = rng.choice(n, size=48, replace=False)
    wi, wd, ws = binding.port().nn2(q[pick], t)
    assert np.array_equal(bi[pick], wi) and np.array_equal(bd[pick], wd) and np.array_equal(sd[pick], ws)


def test_adaptive_tiles_on_ragged_database(m):
    """scan_kernel_adaptive: image sizes on and around every tile boundary (multiples of 64 up to
    1025, so every 256/128/64 remainder combination occurs, forward and in the pruned backward
    pass), all pairs as one batch large enough for the R = 8 regime, against the oracle; the
    fixed-tile kernels (variant bit 18) must give the same lists."""
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 319, 320, 321, 383, 384, 385, 447, 448,
             449, 511, 512, 513, 575, 576, 577, 640, 704, 768, 832, 896, 959, 960, 961, 1023, 1024, 1025]
    db, off = cases.synth.sfm_database(len(sizes), sizes, seed=77, share_frac=0.5, window=40)
    pairs = cases.synth.all_pairs(len(sizes))
    pairs = np.concatenate([pairs, pairs[:, ::-1]])  # both orientations
    m.db_upload(db, off)
    offs, lists = m.match_pairs(pairs, 70, 1.2)
    # fixed 512-query tiles; 512-query adaptive tiles, 1- and 4-wave; forced full-tile sizes 256 / 128 / 64,
    # the last two also with integer keys
    for variant in (0x40000, 0x80000, 0x1000000, 0x40000000, 0x20000000, 0x10000000, 0x20020000, 0x10020000):
        m.set_variant(variant)
        try:
            offs_v, lists_v = m.match_pairs(pairs, 70, 1.2)
        finally:
            m.set_variant(0)
        assert np.array_equal(offs, offs_v) and np.array_equal(lists, lists_v), variant
    assert len(lists) > 20000
    o = binding.port()
    for k in range(0, len(pairs), 7):
        i, j = pairs[k]
        want = o.match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2)
        assert np.array_equal(lists[int(offs[k]):int(offs[k + 1])], want), (k, i, j)


def test_fuzz_random_shapes_and_parameters(m):
    """200 seeded random cases: sizes 0..400 per side, threshold 1..300, ratio 0.5..4, planted or
    dense duplicates -- single-pair call against the oracle, and the same pairs once more through
    the batched call."""
    rng = np.random.default_rng(20251110)
    o = binding.port()
    sets, params, want = [], [], []
    for k in range(200):
        n1, n2 = (int(v) for v in rng.integers(0, 401, 2))
        a, b = cases.synth.orb_like_pair(max(n1, 1), max(n2, 1), seed=1000 + k, match_frac=float(rng.uniform(0, 1)),
                                         dup_frac=float(rng.uniform(0, 0.3)), confuse_frac=float(rng.uniform(0, 0.3)))
        a, b = a[:n1], b[:n2]
        thr, ratio = int(rng.integers(1, 301)), float(rng.uniform(0.5, 4.0))
        w = o.match(a, b, thr, ratio)
        assert np.array_equal(m.match(a, b, thr, ratio), w), (k, n1, n2, thr, ratio)
        sets += [a, b]
        params.append((thr, ratio))
        want.append(w)
    off = np.zeros(len(sets) + 1, np.uint32)
    off[1:] = np.cumsum([len(s) for s in sets])
    m.db_upload(np.concatenate(sets), off)
    pairs = np.arange(400, dtype=np.uint32).reshape(200, 2)
    for thr, ratio in ((70, 1.2), (90, 1.0)):
        offs, lists = m.match_pairs(pairs, thr, ratio)
        for k in range(0, 200, 3):
            assert np.array_equal(lists[int(offs[k]):int(offs[k + 1])], o.match(sets[2 * k], sets[2 * k + 1], thr, ratio)), k


def test_repeated_runs_are_bit_identical(m):
    """Race check: the same batch 25 times over (ragged database, pruned backward pass, atomics in
    nominate_kernel) must give byte-identical offsets and lists every time."""
    rng = np.random.default_rng(9)
    sizes = rng.integers(100, 1400, size=60)
    db, off = cases.synth.sfm_database(60, sizes, seed=91, share_frac=0.5, window=10)
    pairs = cases.synth.all_pairs(60)
    m.db_upload(db, off)
    o0, l0 = m.match_pairs(pairs, 70, 1.2)
    assert len(l0) > 5000
    for _ in range(25):
        o, l = m.match_pairs(pairs, 70, 1.2)
        assert np.array_equal(o, o0) and np.array_equal(l, l0)
