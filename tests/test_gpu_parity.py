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


# Experiment knob (choose_shape, csrc/vsmatch.hip): low byte 1 / 4 / 8 = latency shape (64-query
# tiles, 4-wave workgroups) / fixed 256-query popcount tiles / adaptive tiles; bits 8-15 train
# splits; bit 16 no backward pruning; bit 17 integer keys (popcount); bit 25 popcount everywhere,
# 26 in the pruned backward pass, 27 in the latency shape; bits 28-30 adaptive full-tile size.
POPCOUNT, INT_KEYS, NO_PRUNE, POP_BACK, POP_LAT = 0x2000000, 0x20000, 0x10000, 0x4000000, 0x8000000
TILE256, TILE128, TILE64 = 0x40000000, 0x20000000, 0x10000000
M16, M16_OFF = 0x1000000, 0x80000  # bit 24 / 19: always / never the 16x16x128 forward scan (mfma16_scan.h) where its shape allows it
FLIP = 0x40000  # bit 18: the other operand codes of the matrix-core scan (+-0.5 <-> shift-free zero-rich, mfma_scan.h CODES)
RUN2, RUN3, RUN4, RUN8 = 0x200000, 0x300000, 0x400000, 0x800000  # consecutive jobs per wave of the batched matrix-core scan


@pytest.mark.parametrize("variant", [1, 4, 8, 0x0301, 0x0308, 0x1708, 0x4008, TILE256 | 8, TILE128 | 8, TILE64 | 8, TILE128 | 0x0208,
                                     M16 | 8, M16_OFF | 8, M16 | NO_PRUNE | 8,
                                     FLIP | 1, FLIP | 8, FLIP | 0x0301, FLIP | 0x0308, FLIP | TILE256 | 8, FLIP | TILE128 | 8, FLIP | TILE64 | 8,
                                     POPCOUNT | 1, POPCOUNT | 4, POPCOUNT | 8, POPCOUNT | 0x0308, POPCOUNT | 0x0301,
                                     POPCOUNT | INT_KEYS | 1, POPCOUNT | INT_KEYS | 4, POPCOUNT | INT_KEYS | 8,
                                     POPCOUNT | TILE256 | 8, POPCOUNT | TILE128 | 8, POPCOUNT | TILE64 | 8,
                                     POPCOUNT | INT_KEYS | TILE128 | 0x0308, POP_BACK | 8, POP_LAT | 1, POP_LAT, NO_PRUNE | 8])
def test_pair_all_kernel_shapes(m, variant):
    """Every kernel instantiation of the library -- matrix-core scan at 256/128/64-query adaptive
    tiles and in the 4-wave latency shape, with forced train splits; popcount scan in the same
    shapes with float and integer keys -- gives the recorded reference lists."""
    m.set_variant(variant)
    try:
        for name in ("ex3_thr70_r1.2", "planted_300x777", "planted_1023x1025", "planted_4100x130",
                     "edge_all_identical", "edge_all_256_thr300_ratio1", "edge_tie_best"):
            d1, d2, thr, ratio = cases.build_case(name)
            assert sha(m.match(d1, d2, thr, ratio)) == REF["cases"][name]["sha256"], (hex(variant), name)
    finally:
        m.set_variant(0)


@pytest.mark.parametrize("thr,ratio", [(70, 1.2), (256, 1.0), (300, 1.0), (257, 0.9), (50, 3.0), (100, 0.5), (1, 1.2),
                                       (0, 1.2), (-5, 1.2), (70, 0.0), (70, -1.0), (70, 4.0), (200, 1.01), (70, float("nan")),
                                       (128, 1.5), (255, 1.0)])
def test_accept_limit_of_the_matrix_core_scan(m, thr, ratio):
    """The matrix-core scan starts (best, second) at the smallest distance the accept test
    cannot tell from anything larger (accept_limit, vsmatch.hip); whatever threshold and ratio
    that limit is derived from, the list is the oracle's."""
    for variant in (8, TILE64 | 8, 0x0308, 1, 0x0301, FLIP | 8, FLIP | TILE64 | 8, FLIP | 1, M16 | 8):
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


@pytest.mark.parametrize("variant", [8, TILE64 | 8, TILE128 | 8, 0x0308, 1, 0x0501, POPCOUNT | 8, FLIP | 8, FLIP | TILE128 | 8, FLIP | 1, M16 | 8])
def test_real_descriptors_every_tail_length(m, variant):
    """Real ORB descriptors (many near neighbours, ties and second bests right next to the best)
    at train sizes around the 32-row step of the matrix-core scan -- the partial last step is
    handled apart from the pipelined loop; also raw 2-NN, whose second best is refined for every query."""
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


def _neighbours_next_to_the_best(n, seed, gap):
    """Queries whose two nearest train descriptors sit `gap` rows apart: the matrix-core scan
    keeps one minimum per group of 16 rows (rows 4h + (k & 3) + 8 (k >> 2) of a 32-row step), so
    a second best inside the best's group only comes out of refine_second.  Flip counts vary so
    that the ratio test passes for some queries and fails for others."""
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 2 ** 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    per = 2 + gap
    t = rng.integers(0, 2 ** 63, size=(n * per, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n * per, 4), dtype=np.uint64)

    def flipped(d, k):
        out = d.copy()
        for bit in rng.choice(256, size=k, replace=False):
            out[bit // 64] ^= np.uint64(1) << np.uint64(bit % 64)
        return out

    for i in range(n):
        k1 = int(rng.integers(0, 30))
        k2 = k1 + int(rng.integers(0, 12))  # 0: a tie; small: the ratio test fails; large: it passes
        a = i * per
        b = a + gap
        first, second = (a, b) if rng.integers(0, 2) else (b, a)
        t[first] = flipped(q[i], k1)
        t[second] = flipped(q[i], k2)
    return q, t


@pytest.mark.parametrize("gap", [1, 2, 3, 8, 9, 16, 24, 4, 5, 33])
def test_second_best_next_to_the_best(m, gap):
    """gap 1-3, 8, 9, 16, 24: both neighbours can share a group (same 4-row chunk, or chunks 8
    rows apart); 4, 5: the other half of the step; 33: another step.  Single pair (merge_refine),
    batch (nominate / count_list refinement, both directions), raw 2-NN (finalize), each against
    the oracle, in every scan shape."""
    q, t = _neighbours_next_to_the_best(150, 100 + gap, gap)
    o = binding.port()
    want, want_t = o.match(q, t, 70, 1.2), o.match(t, q, 70, 1.2)
    assert 10 < len(want) < 140  # some pass the ratio test, some do not
    off = np.array([0, len(q), len(q) + len(t)], np.uint32)
    db = np.concatenate([q, t])
    pairs = np.array([[0, 1], [1, 0]] * 1100, np.uint32)  # enough jobs for the adaptive regime
    for variant in (0, 1, 8, TILE64 | 8, 0x0308, 0x0301, NO_PRUNE | 8, POPCOUNT | 8, RUN3 | TILE256 | 8, RUN8 | NO_PRUNE | TILE256 | 8,
                    FLIP, FLIP | 1, FLIP | 8, FLIP | TILE64 | 8, FLIP | RUN3 | TILE256 | 8, M16, M16 | 8, M16 | NO_PRUNE | 8, M16_OFF):
        m.set_variant(variant)
        try:
            assert np.array_equal(m.match(q, t, 70, 1.2), want), (hex(variant), gap)
            assert np.array_equal(m.match(t, q, 70, 1.2), want_t), (hex(variant), gap)
            for g, w in zip(m.nn2(q, t), o.nn2(q, t)):
                assert np.array_equal(g, w), (hex(variant), gap)
            m.db_upload(db, off)
            offs, lists = m.match_pairs(pairs, 70, 1.2)
            for k in (0, 1, 2198, 2199):
                assert np.array_equal(lists[int(offs[k]):int(offs[k + 1])], want if k % 2 == 0 else want_t), (hex(variant), gap, k)
            assert int(offs[-1]) == 1100 * (len(want) + len(want_t))
        finally:
            m.set_variant(0)


def _flip(d, bits):
    d = d.copy()
    for k in bits:
        d[k // 64] ^= np.uint64(1) << np.uint64(k % 64)
    return d


@pytest.mark.parametrize("d", [0, 5, 10, 25, 58])
def test_group_table_certificate_at_its_boundary(m, d):
    """The consumer kernels skip the refinement of a second best when the best row's group table
    entry proves no group mate can matter (vsmatch_kernels.h second_certified: gmin[t] >=
    min_second[d] + d).  The tight case of the triangle inequality: train row t' = t with a set A of g
    bits flipped sits next to t (same group), the query is t with d of A's bits flipped, so that
    dist(q, t') = g - d exactly; g around min_second[d] + d, where the certificate starts to hold and
    the ratio test starts to pass.  Batched both ways (forward and backward results), against the oracle."""
    rng = np.random.default_rng(300 + d)
    need = int(np.ceil(1.2 * d - 1e-9))
    base = cases.synth.random_descriptors(256, 77 + d, 3)
    q, t = base[:96].copy(), cases.synth.random_descriptors(128, 78 + d, 4)
    for k in range(48):
        g = need + d + (k % 7) - 3          # gmin of the pair of rows: three below the bound to three above
        if g <= d:
            continue
        a = rng.choice(256, size=g, replace=False)
        row = 2 * k + (k % 2) * 32          # t' right behind t; some pairs straddle a 4-row chunk
        row = min(row, 126)
        t[row + 1] = _flip(t[row], a)
        q[k] = _flip(t[row], a[:d])
    o = binding.port()
    want, want_t = o.match(q, t, 70, 1.2), o.match(t, q, 70, 1.2)
    assert 8 < len(want) < 48 and 8 < len(want_t) < 48  # some pass, some fail, in both directions
    db = np.concatenate([q, t])
    off = np.array([0, len(q), len(q) + len(t)], np.uint32)
    pairs = np.array([[0, 1], [1, 0]] * 1100, np.uint32)  # enough jobs for the pruned batched path
    for variant in (0, TILE64 | 8, NO_PRUNE | 8, RUN4 | TILE256 | 8):
        m.set_variant(variant)
        try:
            m.db_upload(db, off)
            offs, lists = m.match_pairs(pairs, 70, 1.2)
            for k in (0, 1, 2198, 2199):
                assert np.array_equal(lists[int(offs[k]):int(offs[k + 1])], want if k % 2 == 0 else want_t), (hex(variant), d, k)
        finally:
            m.set_variant(0)


def test_real_like_database_batched():
    """A database with the distance statistics of real ORB descriptors (unrelated descriptors
    126 +- 19 bits apart; duplicates, ties, crowded neighbourhoods): the batched matcher's lists
    equal the oracle's in every form -- matrix cores, without pruning, popcount, mixed."""
    with vsm.Matcher(0) as mm:
        db, off = _real_like_database(64, 900, 5)  # 2016 pairs: enough waves for the batched (adaptive) scan
        pairs = cases.synth.all_pairs(64)
        mm.db_upload(db, off)
        want = [binding.port().match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2) for i, j in pairs.tolist()]
        assert sum(len(w) for w in want) > 10000
        for variant in (0, NO_PRUNE, POPCOUNT, POP_BACK, TILE64, 0x0200, 1, RUN2 | TILE256, RUN3 | TILE256, RUN8 | TILE256, RUN4 | NO_PRUNE | TILE256,
                        FLIP, FLIP | TILE64, FLIP | RUN4 | TILE256, M16, M16_OFF, M16 | NO_PRUNE):
            mm.set_variant(variant)
            offs, out = mm.match_pairs(pairs, 70, 1.2)
            for k, w in enumerate(want):
                assert np.array_equal(out[int(offs[k]):int(offs[k + 1])], w), (hex(variant), k)


@pytest.mark.parametrize("thr,ratio", [(70, 1.2), (1, 1.2), (40, 1.0), (40, 2.5), (100, 1.05), (100, 4.0), (200, 1.2),
                                       (256, 1.0), (300, 0.5), (83, 1.2), (70, 0.0)])
def test_batched_pruned_path_under_many_accept_tests(thr, ratio):
    """The pruned batched path (accept masks by the scan's upper bound, nominate, pruned backward scan,
    refinement after the cross-check with the group-table certificate) under thresholds and ratios
    that move the accept limit, the certificate's bound min_second[d] + d and the table itself:
    real-like database (crowded neighbourhoods, duplicates), 120 sampled pairs against the oracle,
    with and without runs of jobs, and the unpruned path as a second opinion on every pair."""
    with vsm.Matcher(0) as mm:
        db, off = _real_like_database(64, 900, 5)
        pairs = cases.synth.all_pairs(64)
        mm.db_upload(db, off)
        mm.set_variant(8)
        offs, out = mm.match_pairs(pairs, thr, ratio)
        mm.set_variant(RUN4 | TILE256 | 8)
        offs_r, out_r = mm.match_pairs(pairs, thr, ratio)
        mm.set_variant(NO_PRUNE | 8)
        offs_n, out_n = mm.match_pairs(pairs, thr, ratio)
    assert np.array_equal(offs, offs_r) and np.array_equal(out, out_r)
    assert np.array_equal(offs, offs_n) and np.array_equal(out, out_n)
    o = binding.port()
    for k in np.random.default_rng(thr).choice(len(pairs), size=120, replace=False).tolist():
        i, j = pairs[k]
        want = o.match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], thr, ratio)
        assert np.array_equal(out[int(offs[k]):int(offs[k + 1])], want), (thr, ratio, k)


def test_the_16x16_forward_scan_at_every_tile_and_tail_boundary():
    """csrc/mfma16_scan.h forced on (knob bit 24) on a RAGGED real-like database whose image sizes sit on its
    boundaries: 16-query blocks, balanced tiles of at most 224 queries (1 .. 3 tiles, every instantiation 2 .. 14
    blocks), supersteps of 64 train rows and the half-step tail (0 .. 63 rows behind the last superstep), empty
    images.  All pairs, both scan directions of the unpruned path and the pruned path with its accept masks, against
    the 32x32x64 kernel (bit 19) and against the oracle."""
    sizes = [0, 1, 15, 16, 17, 31, 32, 33, 47, 63, 64, 65, 95, 96, 97, 111, 127, 128, 129, 159, 161, 191, 193, 223, 224, 225,
             255, 257, 287, 289, 319, 321, 351, 353, 383, 385, 415, 417, 447, 448, 449, 479, 481, 511, 513, 543, 545, 671, 673, 700]
    rng = np.random.default_rng(77)
    base = np.concatenate(cases.ex3())
    parts = []
    for n in sizes:
        d = base[rng.integers(0, len(base), size=n)].copy()
        d[:, 0] ^= np.uint64(1) << rng.integers(0, 64, size=n).astype(np.uint64)  # near neighbours, a few exact duplicates
        parts.append(d)
    db = np.concatenate(parts)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    pairs = np.array([(i, j) for i in range(len(sizes)) for j in range(len(sizes)) if i != j], np.uint32)
    with vsm.Matcher(0) as mm:
        mm.db_upload(db, off)
        res = {}
        for name, variant in (("m16", M16 | TILE256 | 8), ("m16_unpruned", M16 | NO_PRUNE | TILE256 | 8), ("m32", M16_OFF | TILE256 | 8)):
            mm.set_variant(variant)
            res[name] = mm.match_pairs(pairs, 70, 1.2)
            assert mm.last_scan_kernel(0) == (2 if name == "m32" else 3), name
        mm.set_variant(M16 | TILE256 | 8)  # (256-query tiles forced: small images would otherwise get 128-query tiles, which the 16x16 kernel does not replace)
        loose = mm.match_pairs(pairs, 256, 1.0)  # every query tracked to distance 256
        mm.set_variant(M16_OFF | TILE256 | 8)
        loose32 = mm.match_pairs(pairs, 256, 1.0)
    for name in ("m16_unpruned", "m32"):
        assert np.array_equal(res["m16"][0], res[name][0]) and np.array_equal(res["m16"][1], res[name][1]), name
    assert np.array_equal(loose[0], loose32[0]) and np.array_equal(loose[1], loose32[1])
    o = binding.port()
    offs, out = res["m16"]
    assert int(offs[-1]) > 5000
    for k in rng.choice(len(pairs), size=400, replace=False).tolist():
        i, j = pairs[k]
        want = o.match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2)
        assert np.array_equal(out[int(offs[k]):int(offs[k + 1])], want), (sizes[i], sizes[j])
        want = o.match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 256, 1.0)
        assert np.array_equal(loose[1][int(loose[0][k]):int(loose[0][k + 1])], want), (sizes[i], sizes[j])


def test_consumer_kernels_at_list_length_boundaries():
    """nominate_kernel / count_list_kernel / fill_list_kernel (vsmatch_kernels.h) on pairs whose number of forward-accepted
    rows sits on their boundaries: 256-entry passes, count_list's turns of 512 entries (two per thread), nominate's
    2048-entry list in LDS -- K planted matches per pair of 2304-descriptor images, a third of them with a group mate
    close enough that the group table does not certify the second best (count_list's refinement teams decide, some
    of them against the match).  Both orientations, the pruned path on both matrix-core shapes and after a popcount
    scan (nominate then lists from the results, not from accept bits), the unpruned path, against the oracle."""
    n = 2304
    counts = [0, 1, 255, 256, 257, 511, 512, 513, 767, 768, 769, 1023, 1024, 1025, 2047, 2048, 2049, 2304]
    rng = np.random.default_rng(2024)
    parts, pairs = [], []
    for k_img, k in enumerate(counts):
        a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4)).astype(np.uint64)
        b = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4)).astype(np.uint64)
        rows_a = rng.choice(n, size=k, replace=False)
        rows_b = rng.choice(n, size=k, replace=False)
        for ra, rb in zip(rows_a.tolist(), rows_b.tolist()):
            d = a[ra].copy()
            for bit in rng.choice(256, size=int(rng.integers(0, 4)), replace=False).tolist():
                d[bit >> 6] ^= np.uint64(1) << np.uint64(bit & 63)
            b[rb] = d
        taken = set(rows_b.tolist())
        for t, (ra, rb) in enumerate(list(zip(rows_a.tolist(), rows_b.tolist()))[::3]):  # a group mate 0 .. 5 or 2 .. 40 bits from the copy
            mate = rb ^ 1
            if mate in taken or mate >= n:
                continue
            taken.add(mate)
            d = b[rb].copy()
            for bit in rng.choice(256, size=int(rng.integers(0, 6) if t & 1 else rng.integers(2, 41)), replace=False).tolist():
                d[bit >> 6] ^= np.uint64(1) << np.uint64(bit & 63)
            b[mate] = d
        parts += [a, b]
        pairs += [(2 * k_img, 2 * k_img + 1), (2 * k_img + 1, 2 * k_img)]
    db = np.ascontiguousarray(np.concatenate(parts))
    off = (np.arange(2 * len(counts) + 1) * n).astype(np.uint32)
    pairs = np.array(pairs, np.uint32)
    o = binding.port()
    want = [o.match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2) for i, j in pairs.tolist()]
    lens = [len(w) for w in want]
    assert lens[0] == 0 and lens[1] == 0 and max(lens) > 1500
    assert any(l < k for l, k in zip(lens[::2], counts))  # some planted matches fall to their close group mate
    with vsm.Matcher(0) as mm:
        mm.db_upload(db, off)
        for variant in (8, M16_OFF | TILE256 | 8, POPCOUNT | 8, NO_PRUNE | 8, FLIP | M16_OFF | 8):
            mm.set_variant(variant)
            offs, out = mm.match_pairs(pairs, 70, 1.2)
            if variant == 8:
                assert mm.last_scan_kernel(0) == 3  # 2304 descriptors per image: the 16x16x128 forward scan
            for k, w in enumerate(want):
                assert np.array_equal(out[int(offs[k]):int(offs[k + 1])], w), (hex(variant), counts[k // 2], k & 1)


def test_nn2_vs_oracle(m):
    for variant in (0, 8, 1, TILE64 | 8, TILE128 | 8, POPCOUNT | 8, POPCOUNT | INT_KEYS, POP_LAT | 1, FLIP | 8, FLIP | TILE64 | 8, FLIP | 1, M16 | 8):
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
        m.set_variant(NO_PRUNE)
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
    for devices in (None, "0,0"):  # one thread-local context; two contexts of vsm_match_pairs_multi on the one GPU
        env = dict(os.environ)
        env.pop("VSM_DEVICES", None)
        if devices:
            env["VSM_DEVICES"] = devices
        r = subprocess.run([exe, os.path.join(g, "ex3_kd0.npy"), os.path.join(g, "ex3_kd1.npy"),
                            os.path.join(g, "ex3_matches.npy")], capture_output=True, text=True, timeout=100, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert ("VSM_DEVICES" in r.stdout) == bool(devices)


def test_multi_context_batch_equals_single_context(m):
    """vsm_db_upload_multi / vsm_match_pairs_multi (match_all on several GPUs from one process,
    src/sfm.cpp:1276-1321): two and three contexts on device 0 give the offsets and lists of the
    single-context call, in the order of the pair list -- ragged database, more pairs than one
    4096-pair block per context, both orientations, an empty image; capacity and bad-index errors
    behave like vsm_match_pairs."""
    rng = np.random.default_rng(17)
    sizes = rng.integers(0, 500, size=150)
    sizes[7] = 0
    db, off = cases.synth.sfm_database(150, sizes, seed=23, share_frac=0.5, window=10)
    pairs = cases.synth.all_pairs(150)
    pairs = np.concatenate([pairs, pairs[::-1, ::-1]])  # 22350 pairs: 6 blocks
    m.db_upload(db, off)
    want_off, want = m.match_pairs(pairs, 70, 1.2)
    assert len(want) > 20000
    for devices in ([0, 0], [0, 0, 0], [0]):
        with vsm.MultiMatcher(devices) as mm:
            mm.db_upload(db, off)
            got_off, got = mm.match_pairs(pairs, 70, 1.2)
            assert np.array_equal(got_off, want_off) and np.array_equal(got, want), devices
            o2, l2 = mm.match_pairs(pairs[:5000], 90, 1.0)  # another batch on the same contexts
            w2 = m.match_pairs(pairs[:5000], 90, 1.0)
            assert np.array_equal(o2, w2[0]) and np.array_equal(l2, w2[1])
            o3, l3 = mm.match_pairs(np.zeros((0, 2), np.uint32))
            assert list(o3) == [0] and len(l3) == 0
            with pytest.raises(vsm.VsmError) as e:
                mm.match_pairs(pairs, 70, 1.2, cap=100)
            assert e.value.code == vsm.matcher.VSM_ERR_CAPACITY
            with pytest.raises(vsm.VsmError):
                mm.match_pairs(np.array([[0, 999]], np.uint32), cap=10)


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
    """One-direction scan at 200k x 200k (several train splits): invariants on every query,
    exact values on a sample."""
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
    """Adaptive tiles: image sizes on and around every tile boundary (multiples of 64 up to
    1025, so every 256/128/64 remainder combination occurs, forward and in the pruned backward
    pass), all pairs as one batch large enough for the adaptive regime, against the oracle; every
    tile size and scan form must give the same lists."""
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 319, 320, 321, 383, 384, 385, 447, 448,
             449, 511, 512, 513, 575, 576, 577, 640, 704, 768, 832, 896, 959, 960, 961, 1023, 1024, 1025]
    db, off = cases.synth.sfm_database(len(sizes), sizes, seed=77, share_frac=0.5, window=40)
    pairs = cases.synth.all_pairs(len(sizes))
    pairs = np.concatenate([pairs, pairs[:, ::-1]])  # both orientations
    m.db_upload(db, off)
    offs, lists = m.match_pairs(pairs, 70, 1.2)
    # forced full-tile sizes 256 / 128 / 64 on the matrix cores and with the popcount scan (float and
    # integer keys), no pruning, popcount backward pass only
    # runs of 2 / 3 / 8 consecutive jobs per wave (ragged images: a run ends where the query image changes)
    for variant in (TILE256, TILE128, TILE64, POPCOUNT | TILE256, POPCOUNT | TILE128, POPCOUNT | INT_KEYS | TILE64,
                    NO_PRUNE, POP_BACK, RUN2 | TILE256, RUN3 | TILE256, RUN8 | TILE256, RUN8 | NO_PRUNE | TILE256):
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
