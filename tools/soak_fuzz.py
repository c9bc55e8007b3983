#!/usr/bin/env python3
"""Randomised differential soak of the batched matcher against the oracle (GPU box, through gpurun).

    python3 tools/soak_fuzz.py [seconds=240] [seed=1] [single | frontend | landmarks | bow]

Each case draws a database (uniform / ragged image sizes incl. empty images; planted overlap,
real-ORB statistics or heavy duplicates), an accept test (threshold, ratio), a pair list (random
ordered pairs, both orientations, repeats, self pairs) and an experiment-knob setting, runs
vsm_match_pairs and compares EVERY pair's list with the oracle port (16 threads over pairs).
Prints one line per case and a summary; exits 1 at the first difference, naming the case.
With `single`: vsm_match_pair and vsm_nn2 on random pairs of sets instead (latency shapes, train splits).
With `frontend`: vsm_detect_and_describe_batch on random image sets against the oracle's detector + ORB.
With `landmarks` / `bow`: vsm_match_landmarks / the BoW database entry points against their oracles.
Not a test (the -m gpu suite holds the fixed cases): a tool for hunting rare cases after a change
to the scan, the pruned backward pass or the refinement.
"""
import importlib
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding  # noqa: E402

synth = importlib.import_module("visual-slam_amd.synth")
matcher = importlib.import_module("visual-slam_amd.matcher")

TILE256, TILE128, TILE64 = 0x40000000, 0x20000000, 0x10000000
POPCOUNT, POP_BACK, NO_PRUNE, INT_KEYS = 1 << 25, 1 << 26, 1 << 16, 1 << 17
RUN = lambda j: j << 20  # noqa: E731
M16, M16_OFF, FLIP = 0x1000000, 0x80000, 0x40000  # round 3: always / never the 16x16x128 forward scan; the other operand codes
VARIANTS = [0, 0, 0, 0, TILE256, TILE128, TILE64, NO_PRUNE, POP_BACK, POPCOUNT | TILE256, POPCOUNT | INT_KEYS | TILE64,
            M16 | TILE256, M16 | TILE256, M16 | TILE256 | 8, M16 | NO_PRUNE | TILE256, M16_OFF, FLIP, FLIP | TILE128, FLIP | TILE64,
            RUN(2) | TILE256, RUN(3) | TILE256, RUN(4), RUN(8) | TILE256, RUN(8) | NO_PRUNE | TILE256, 0x0208, 0x0308, 8, 4, 1]


def real_pool():
    g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    return np.concatenate([np.load(os.path.join(g, "ex3_kd0.npy")), np.load(os.path.join(g, "ex3_kd1.npy"))])


def make_database(rng, real, case):
    n_img = int(rng.integers(2, 41))
    kind = rng.choice(["planted", "real", "dups"], p=[0.45, 0.4, 0.15])
    if rng.random() < 0.4:
        sizes = np.full(n_img, int(rng.choice([1, 31, 32, 33, 64, 200, 333, 500, 777, 1000, 1500, 2000])))
    else:
        hi = int(rng.choice([70, 300, 1100, 2600]))
        sizes = rng.integers(0, hi + 1, size=n_img)
        sizes[rng.random(n_img) < 0.08] = 0
    sizes = np.minimum(sizes, max(1, 30000 // n_img)).astype(np.int64)
    off = np.zeros(n_img + 1, np.uint32)
    off[1:] = np.cumsum(sizes)
    total = int(off[-1])
    if kind == "planted" and sizes.min() > 0:
        db, off = synth.sfm_database(n_img, [int(s) for s in sizes], seed=10_000 + case,
                                     share_frac=float(rng.uniform(0.1, 0.9)), window=int(rng.integers(1, n_img + 1)))
    elif kind == "dups":
        base = synth.random_descriptors(max(1, total // 6 + 1), seed=20_000 + case)
        db = base[rng.integers(0, len(base), size=total)] ^ synth.flip_masks(max(total, 1), 3, case, levels=(7, 8))[:total]
        db[rng.random(total) < 0.5] = base[0]
    else:
        kind = "real" if kind != "planted" else "real(ragged)"
        flips = synth.flip_masks(max(total, 1), 5, case, levels=(int(rng.integers(4, 7)),))[:total]
        db = real[rng.integers(0, len(real), size=total)] ^ flips
    return np.ascontiguousarray(db, dtype=np.uint64).reshape(-1, 4), off.astype(np.uint32), kind


def single_pairs(budget, seed):
    """vsm_match_pair and vsm_nn2 (host pointers; the latency shapes and train splits) on random pairs of sets."""
    rng = np.random.default_rng(seed)
    real = real_pool()
    m = matcher.Matcher(0)
    o = binding.port()
    variants = [0, 0, 0, 1, 4, 8, 0x0301, 0x0308, 0x0208, 0x1708, POPCOUNT | 8, POPCOUNT | 1, TILE64 | 8, TILE128 | 8, TILE256 | 8,
                M16 | 8, M16 | TILE256 | 8, FLIP | 8, FLIP | 1]
    t_end = time.time() + budget
    case = 0
    while time.time() < t_end:
        case += 1
        hi = int(rng.choice([40, 300, 2100, 6000]))
        n1, n2 = (int(v) for v in rng.integers(0, hi + 1, 2))
        kind = rng.choice(["planted", "real", "dups"])
        if kind == "planted":
            a, b = synth.orb_like_pair(max(n1, 1), max(n2, 1), seed=seed * 100_000 + case, match_frac=float(rng.uniform(0, 1)),
                                       dup_frac=float(rng.uniform(0, 0.3)), confuse_frac=float(rng.uniform(0, 0.3)))
        elif kind == "real":
            lv = (int(rng.integers(4, 7)),)
            a = real[rng.integers(0, len(real), max(n1, 1))] ^ synth.flip_masks(max(n1, 1), 5, case, levels=lv)
            b = real[rng.integers(0, len(real), max(n2, 1))] ^ synth.flip_masks(max(n2, 1), 6, case, levels=lv)
        else:
            base = synth.random_descriptors(8, seed=case)
            a = base[rng.integers(0, 8, max(n1, 1))] ^ synth.flip_masks(max(n1, 1), 7, case, levels=(8,))
            b = base[rng.integers(0, 8, max(n2, 1))] ^ synth.flip_masks(max(n2, 1), 8, case, levels=(8,))
        a, b = np.ascontiguousarray(a[:n1]), np.ascontiguousarray(b[:n2])
        thr, ratio = (70, 1.2) if rng.random() < 0.5 else (int(rng.integers(1, 301)), float(rng.uniform(0.5, 3.0)))
        variant = int(rng.choice(variants))
        m.set_variant(variant)
        try:
            got = m.match(a, b, thr, ratio)
            nn = m.nn2(a, b) if n1 and n2 and rng.random() < 0.5 else None
        finally:
            m.set_variant(0)
        ok = np.array_equal(got, o.match(a, b, thr, ratio))
        if ok and nn is not None:
            ok = all(np.array_equal(g, w) for g, w in zip(nn, o.nn2(a, b)))
        line = f"case {case} seed {seed}: {kind} {n1} x {n2}, thr {thr} ratio {ratio:.4f}, variant {variant:#x}: {len(got)} matches"
        if not ok:
            print(line + " -- DIFFERENT", flush=True)
            return 1
        if case % 200 == 0:
            print(line, flush=True)
    print(f"soak (single pairs): {case} cases, all equal to the oracle", flush=True)
    return 0


def random_image(rng, w, h):
    kind = rng.choice(["texture", "noise", "blocks", "gradient", "flat", "checker"], p=[0.35, 0.15, 0.2, 0.1, 0.05, 0.15])
    if kind == "noise":
        return rng.integers(0, 256, (h, w)).astype(np.uint8)
    if kind == "flat":
        return np.full((h, w), int(rng.integers(0, 256)), np.uint8)
    if kind == "gradient":
        yy, xx = np.indices((h, w))
        return ((xx * int(rng.integers(1, 4)) + yy * int(rng.integers(0, 3))) % 256).astype(np.uint8)
    if kind == "checker":  # exactly equal eigenvalues: the tie order of the candidates matters
        yy, xx = np.indices((h, w))
        c = int(rng.integers(3, 12))
        return ((yy // c + xx // c) % 2 * int(rng.integers(60, 256))).astype(np.uint8)
    c = int(rng.integers(2, 9))
    a = rng.integers(0, 256, size=(h // c + 2, w // c + 2)).astype(np.float64)
    a = np.kron(a, np.ones((c, c)))[:h, :w]
    if kind == "texture":
        a = (a + np.roll(a, 1, 0) + np.roll(a, 1, 1) + rng.integers(0, 40, size=(h, w))) / 3.2
    return np.clip(a, 0, 255).astype(np.uint8)


def front_end(budget, seed):
    """vsm_detect_and_describe_batch on random image sets against the oracle's detector + ORB, image by image."""
    rng = np.random.default_rng(seed)
    m = matcher.Matcher(0)
    o = binding.port()
    pool = ThreadPoolExecutor(16)
    t_end = time.time() + budget
    case = n_images = n_kp = 0
    while time.time() < t_end:
        case += 1
        w, h = int(rng.integers(40, 801)), int(rng.integers(40, 601))
        n_img = int(rng.integers(1, 13))
        pad = int(rng.integers(0, 9))
        imgs = [random_image(rng, w + pad, h)[:, :w] if pad else random_image(rng, w, h) for _ in range(n_img)]
        nf = int(rng.choice([0, 1, 50, 500, 1500, 4000]))
        rot = bool(rng.random() < 0.8)
        if rng.random() < 0.3:
            os.environ["VSM_BATCH_CHUNK"] = str(int(rng.integers(1, 5)))
        try:
            if rng.random() < 0.3:  # the single-image entry point, image by image
                parts = [m.detect_and_describe(im, nf, rot) for im in imgs]
                off = np.zeros(n_img + 1, np.int64)
                off[1:] = np.cumsum([len(p[0]) for p in parts])
                xy, ang, desc = (np.concatenate([p[k] for p in parts]) for k in range(3))
            else:
                off, xy, ang, desc = m.detect_and_describe_batch(imgs, nf, rot)
        finally:
            os.environ.pop("VSM_BATCH_CHUNK", None)

        def check(k):
            im = np.ascontiguousarray(imgs[k])
            oxy = o.detect_keypoints(im, nf)
            oa, od = o.orb_describe(im, oxy, rot)
            sl = slice(int(off[k]), int(off[k + 1]))
            return np.array_equal(xy[sl], oxy) and np.array_equal(ang[sl], oa) and np.array_equal(desc[sl], od)

        ok = list(pool.map(check, range(n_img)))
        n_images += n_img
        n_kp += int(off[-1])
        line = f"case {case} seed {seed}: {n_img} images {w} x {h} (pitch {w + pad}), num_features {nf}, rotate {rot}: {int(off[-1])} keypoints"
        if not all(ok):
            print(line + f" -- DIFFERENT at images {[k for k, v in enumerate(ok) if not v]}", flush=True)
            return 1
        if case % 20 == 0:
            print(line, flush=True)
    print(f"soak (front end): {case} cases, {n_images} images, {n_kp} keypoints, all equal to the oracle", flush=True)
    return 0


def landmarks(budget, seed):
    """vsm_match_landmarks (find_matches_landmarks, vo_utils.h:95-187) on random frames against the oracle."""
    rng = np.random.default_rng(seed)
    m = matcher.Matcher(0)
    o = binding.port()
    t_end = time.time() + budget
    case = n_matches = 0
    while time.time() < t_end:
        case += 1
        nk, npj = int(rng.integers(0, 2500)), int(rng.integers(0, 4000))
        radius = float(rng.choice([0.0, 3.0, 5.0, 20.0, 20.5, 50.0, 1000.0]))
        f = synth.landmark_frame(nk, npj, seed * 100_000 + case, radius=max(radius, 1.0), max_obs=int(rng.integers(1, 12)),
                                 lattice=bool(rng.random() < 0.4))
        thr, ratio = (70, 1.2) if rng.random() < 0.5 else (int(rng.integers(1, 301)), float(rng.uniform(0.5, 3.0)))
        got = m.match_landmarks(*f, radius, thr, ratio)
        want = o.match_landmarks(*f, radius, thr, ratio)
        n_matches += len(want)
        line = f"case {case} seed {seed}: {nk} keypoints, {npj} landmarks, radius {radius}, thr {thr} ratio {ratio:.4f}: {len(want)} matches"
        if not np.array_equal(got, want):
            print(line + " -- DIFFERENT", flush=True)
            return 1
        if case % 200 == 0:
            print(line, flush=True)
    print(f"soak (landmarks): {case} cases, {n_matches} matches, all equal to the oracle", flush=True)
    return 0


def bow_database(budget, seed):
    """vsm_bowdb_upload / vsm_bowdb_query / vsm_bow_select_pairs on random BoW vectors against the oracle's
    BowDatabase (scores bit for bit, order by (score, image index): tests/test_gpu_bowdb.py::check_query)."""
    from tests.test_gpu_bowdb import check_query, oracle_db
    rng = np.random.default_rng(seed)
    m = matcher.Matcher(0)
    t_end = time.time() + budget
    case = n_q = 0
    while time.time() < t_end:
        case += 1
        n_img = int(rng.integers(1, 400))
        n_words = int(rng.choice([5, 50, 2000, 100000]))
        per = int(rng.integers(1, min(n_words, 300) + 1))
        off, words, values = synth.bow_vectors(n_img, n_words, per, seed=seed * 100_000 + case, window=int(rng.integers(1, 12)),
                                               share_frac=float(rng.uniform(0, 1)))
        coarse = bool(rng.random() < 0.4)
        if coarse:  # coarse values: equal scores everywhere
            values = np.round(values * 8) / 8 + 0.125
        vecs = [(words[off[i]:off[i + 1]], values[off[i]:off[i + 1]]) for i in range(n_img)]
        m.bowdb_upload((off, words, values))
        k = int(rng.choice([1, 4, 25, 64, 65, 200]))
        qs = rng.integers(0, n_img, size=min(n_img, 12))
        lim = rng.integers(0, n_img + 1, size=len(qs)) if rng.random() < 0.5 else None
        res = m.bowdb_query([list(zip(vecs[q][0].tolist(), vecs[q][1].tolist())) for q in qs], k, db_limit=lim)
        for qi, (q, (idx, score)) in enumerate(zip(qs, res)):
            n_db = n_img if lim is None else int(lim[qi])
            try:
                check_query(oracle_db(vecs, n_db), n_db, vecs[q], k, idx, score)
            except AssertionError:
                print(f"case {case} seed {seed}: {n_img} images, {n_words} words, {per} per image, k {k}, query {q}, limit {n_db} -- DIFFERENT", flush=True)
                raise
        n_q += len(qs)
        # the pair selection of match_bow (sfm.cpp:1366-1392): image i against the images before it, stereo partners
        # skipped.  Which of several EQUAL scores make the cut is open in the reference (std::partial_sort; one
        # shared word with the smaller weight on the query's side gives equal scores), so the list is checked
        # against the library's own queries with limit i -- which check_query above holds to the oracle
        nc = int(rng.choice([1, 7, 25]))
        frames = np.arange(n_img) // 2
        got = m.bow_select_pairs(frames, nc)
        allq = m.bowdb_query((off, words, values), nc, db_limit=np.arange(n_img))
        want = [(i, int(j)) for i, (idx, _) in enumerate(allq) for j in idx if frames[i] != frames[int(j)]]
        if [tuple(p) for p in got.tolist()] != want:
            print(f"case {case} seed {seed}: {n_img} images, {n_words} words, {per} per image: select_pairs({nc}) -- DIFFERENT", flush=True)
            return 1
        for i in rng.integers(0, n_img, size=4):
            check_query(oracle_db(vecs, int(i)), int(i), vecs[i], nc, *allq[i])
        if case % 50 == 0:
            print(f"case {case} seed {seed}: {n_img} images, {n_words} words, {per} per image, k {k}: ok, {len(got)} pairs selected", flush=True)
    print(f"soak (BoW database): {case} cases, {n_q} queries, all equal to the oracle", flush=True)
    return 0


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if len(sys.argv) > 3 and sys.argv[3] == "landmarks":
        return landmarks(budget, seed)
    if len(sys.argv) > 3 and sys.argv[3] == "bow":
        return bow_database(budget, seed)
    if len(sys.argv) > 3 and sys.argv[3] == "frontend":
        return front_end(budget, seed)
    if len(sys.argv) > 3 and sys.argv[3] == "single":
        return single_pairs(budget, seed)
    rng = np.random.default_rng(seed)
    real = real_pool()
    m = matcher.Matcher(0)
    o = binding.port()
    pool = ThreadPoolExecutor(16)
    t_end = time.time() + budget
    case = n_pairs_total = n_matches_total = 0
    while time.time() < t_end:
        case += 1
        db, off, kind = make_database(rng, real, seed * 100_000 + case)
        n_img = len(off) - 1
        if len(db) == 0:
            continue
        n_pairs = int(rng.integers(1, 400))
        pairs = rng.integers(0, n_img, size=(n_pairs, 2)).astype(np.uint32)
        if rng.random() < 0.5:
            pairs = np.concatenate([pairs, pairs[: n_pairs // 2, ::-1]])
        if rng.random() < 0.3:  # the all-pairs order of sfm.cpp:1285-1290: runs of jobs with one query image
            pairs = synth.all_pairs(n_img)[: 2000]
        thr, ratio = (70, 1.2) if rng.random() < 0.5 else (int(rng.integers(1, 301)), float(rng.uniform(0.5, 3.0)))
        variant = int(rng.choice(VARIANTS))
        m.db_upload(db, off)
        m.set_variant(variant)
        try:
            offs, lists = m.match_pairs(pairs, thr, ratio)
        finally:
            m.set_variant(0)
        assert int(offs[0]) == 0 and int(offs[-1]) == len(lists)

        def check(k):
            i, j = int(pairs[k, 0]), int(pairs[k, 1])
            out = np.empty((max(1, min(int(off[i + 1] - off[i]), int(off[j + 1] - off[j]))), 2), np.int32)
            n = o.match_db_pair(db, off, i, j, thr, ratio, out)
            return np.array_equal(lists[int(offs[k]):int(offs[k + 1])], out[:n])

        ok = list(pool.map(check, range(len(pairs))))
        n_pairs_total += len(pairs)
        n_matches_total += len(lists)
        line = (f"case {case} seed {seed}: {kind} {n_img} images, {len(db)} descriptors, {len(pairs)} pairs, thr {thr} "
                f"ratio {ratio:.4f}, variant {variant:#x}: {len(lists)} matches")
        if not all(ok):
            bad = [k for k, v in enumerate(ok) if not v]
            print(line + f" -- DIFFERENT at pairs {bad[:10]} ({[pairs[k].tolist() for k in bad[:10]]})", flush=True)
            return 1
        if case % 20 == 0:
            print(line, flush=True)
    print(f"soak: {case} cases, {n_pairs_total} pairs, {n_matches_total} matches, all equal to the oracle", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
