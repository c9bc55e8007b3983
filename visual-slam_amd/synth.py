"""Deterministic synthetic ORB-like descriptor sets.

Uniform-random 256-bit descriptors give *zero* matches at the reference's
default threshold 70 (the minimum of ~2000 Binomial(256, 1/2) draws is ~100),
so every set made here carries planted structure: noisy re-observations of
shared "landmarks" (true matches at Hamming distance ~10-60), exact duplicates
inside a set (lowest-index tie-breaking, keypoints.h:273), near-duplicates
inside a set (ratio-test failures, keypoints.h:296) and fresh random
descriptors (threshold failures).

Everything is a pure function of (sizes, seed) built on a counter-based
splitmix64, so the CPU oracle, the GPU path and every rank of a multi-GPU run
regenerate identical bytes without shipping data.

Descriptor layout: ``uint64[n, 4]``, bit k of the reference's
``std::bitset<256>`` (keypoints.h:53) = bit ``k % 64`` of word ``k // 64``.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def _key(seed: int, stream: int) -> np.uint64:
    with np.errstate(over="ignore"):
        k = np.array([seed & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
        s = np.array([stream & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
        return _mix(_mix(k + _GOLDEN) ^ (s * _M2 + _GOLDEN))[0]


def u64(seed: int, stream: int, shape) -> np.ndarray:
    """Counter-based splitmix64: element i = mix(key(seed, stream) + (i+1)*golden)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        ctr = np.arange(1, n + 1, dtype=np.uint64) * _GOLDEN + _key(seed, stream)
        return _mix(ctr).reshape(shape)


def random_descriptors(n: int, seed: int, stream: int = 0) -> np.ndarray:
    """n iid Bernoulli(1/2) 256-bit descriptors."""
    return u64(seed, 0x1000 + stream, (n, 4))


def flip_masks(n: int, seed: int, stream: int, levels=(3, 4, 5)) -> np.ndarray:
    """XOR masks with per-descriptor bit density 2**-m, m drawn from `levels`
    (AND of m random words): m=3 -> ~32 flipped bits, 4 -> ~16, 5 -> ~8."""
    lv = np.asarray(levels, dtype=np.int64)
    pick = lv[(u64(seed, 0x2000 + stream, (n,)) % np.uint64(len(lv))).astype(np.int64)]
    mask = np.full((n, 4), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    for k in range(int(lv.max())):
        r = u64(seed, 0x3000 + 64 * stream + k, (n, 4))
        use = (pick > k)[:, None]
        mask = np.where(use, mask & r, mask)
    return mask


def _permutation(n: int, seed: int, stream: int) -> np.ndarray:
    return np.argsort(u64(seed, 0x4000 + stream, (n,)), kind="stable")


def _plant_intra(desc: np.ndarray, seed: int, stream: int, dup_frac: float,
                 confuse_frac: float) -> None:
    """In place: exact duplicates and near-duplicates inside one set."""
    n = desc.shape[0]
    if n < 4:
        return
    n_dup = int(n * dup_frac)
    n_con = int(n * confuse_frac)
    pos = _permutation(n, seed, 0x100 + stream)
    src = pos[: n_dup + n_con]
    dst = pos[n_dup + n_con: 2 * (n_dup + n_con)]
    m = min(len(src), len(dst))
    src, dst = src[:m], dst[:m]
    noise = flip_masks(m, seed, 0x200 + stream, levels=(5, 6))  # ~8 / ~4 bits
    noise[: min(n_dup, m)] = 0  # the first n_dup copies are exact duplicates
    desc[dst] = desc[src] ^ noise


def orb_like_pair(n1: int, n2: int, seed: int, match_frac: float = 0.7,
                  dup_frac: float = 0.02, confuse_frac: float = 0.02):
    """A stereo-like pair: set 2 holds noisy copies of
    `match_frac` of set 1 (shuffled), the rest fresh; both sets carry
    intra-set duplicates and near-duplicates."""
    a = random_descriptors(n1, seed, 1)
    _plant_intra(a, seed, 1, dup_frac, confuse_frac)
    b = random_descriptors(n2, seed, 2)
    n_m = int(min(n1, n2) * match_frac)
    src = _permutation(n1, seed, 3)[:n_m]
    dst = _permutation(n2, seed, 4)[:n_m]
    b[dst] = a[src] ^ flip_masks(n_m, seed, 5)
    _plant_intra(b, seed, 2, dup_frac, confuse_frac)
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def planted_sets(n: int, seed: int, planted_frac: float = 0.5):
    """Config 4 (1M x 1M): set B = shuffled A with `planted_frac` noisy copies,
    the rest fresh random."""
    return orb_like_pair(n, n, seed, match_frac=planted_frac, dup_frac=0.001,
                         confuse_frac=0.001)


def sfm_database(n_img: int, n_desc, seed: int, share_frac: float = 0.4,
                 window: int = 5, dup_frac: float = 0.01, confuse_frac: float = 0.01):
    """A database of `n_img` images for all-pairs SfM matching (configs 3, 5).

    Image i re-observes a sliding window of global landmarks so that images
    whose ids differ by k < `window` share about share_frac*(1 - k/window) of
    their descriptors (noisy copies); the rest are fresh.  `n_desc` is an int
    (every image that size) or a sequence of per-image sizes (ragged database).

    Returns (desc uint64[total, 4], offsets uint32[n_img + 1]).
    """
    sizes = np.full(n_img, n_desc, dtype=np.int64) if np.isscalar(n_desc) \
        else np.asarray(n_desc, dtype=np.int64)
    assert sizes.shape == (n_img,)
    offsets = np.zeros(n_img + 1, dtype=np.uint32)
    offsets[1:] = np.cumsum(sizes)
    total = int(offsets[-1])
    desc = random_descriptors(total, seed, 10)
    n_max = int(sizes.max()) if n_img else 0
    n_shared_max = int(n_max * share_frac)
    stride = max(1, n_shared_max // max(1, window))
    n_landmarks = stride * (n_img + window + 1) + n_shared_max
    landmarks = random_descriptors(n_landmarks, seed, 11)
    noise_all = flip_masks(total, seed, 12)
    for i in range(n_img):
        lo, hi = int(offsets[i]), int(offsets[i + 1])
        n = hi - lo
        n_sh = min(int(n * share_frac), n_shared_max)
        if n_sh > 0:
            slot = _permutation(n, seed, 0x10000 + i)[:n_sh]
            lm = i * stride + np.arange(n_sh)
            desc[lo + slot] = landmarks[lm] ^ noise_all[lo + slot]
        view = desc[lo:hi]
        _plant_intra(view, seed, 0x20000 + i, dup_frac, confuse_frac)
    return np.ascontiguousarray(desc), offsets


def all_pairs(n_img: int, frame_of=None) -> np.ndarray:
    """Pair enumeration of src/sfm.cpp:1285-1290: (i, j) for j < i, skipping
    pairs of the same frame (stereo pairs).  `frame_of[i]` is image i's frame
    id; None = every image its own frame.  Returns uint32[n_pairs, 2]."""
    i, j = np.tril_indices(n_img, k=-1)
    if frame_of is not None:
        f = np.asarray(frame_of)
        keep = f[i] != f[j]
        i, j = i[keep], j[keep]
    return np.ascontiguousarray(np.stack([i, j], axis=1).astype(np.uint32))


def _unit(seed: int, stream: int, shape) -> np.ndarray:
    """Uniform doubles in [0, 1) with 53 random bits."""
    return (u64(seed, 0x5000 + stream, shape) >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def landmark_frame(nk: int, n_proj: int, seed: int, width: float = 752.0, height: float = 480.0,
                   radius: float = 20.0, max_obs: int = 7, lattice: bool = False):
    """One odometry frame for find_matches_landmarks (vo_utils.h:95-187).

    `nk` detected keypoints and `n_proj` projected landmarks in a width x height image.
    About 60 % of the landmarks re-observe a keypoint: they project 0 .. 1.3*radius away
    from it (so some fall just outside the gate) and their 1..max_obs observation
    descriptors are noisy copies of that keypoint's descriptor; 10 % are "confusers" that
    project next to an already used keypoint with a slightly worse descriptor (ratio-test
    failures, vo_utils.h:179); 5 % have no observations at all; the rest are unrelated.
    `lattice=True` snaps every coordinate to the integer grid, which puts many
    (keypoint, landmark) distances EXACTLY on an integer radius (3-4-5 triangles), the
    boundary of the `>` in vo_utils.h:138.

    Returns (kp_desc uint64[nk,4], kp_xy f64[nk,2], proj_xy f64[n_proj,2],
             obs_off uint32[n_proj+1], obs_desc uint64[*,4]).
    """
    kp_desc = random_descriptors(nk, seed, 20)
    _plant_intra(kp_desc, seed, 20, 0.02, 0.02)
    kp_xy = _unit(seed, 1, (nk, 2)) * np.array([width, height])
    proj_xy = _unit(seed, 2, (n_proj, 2)) * np.array([width, height])
    kind = _unit(seed, 3, (n_proj,))
    n_obs = 1 + (u64(seed, 0x6000, (n_proj,)) % np.uint64(max(1, max_obs))).astype(np.int64)
    n_obs[(kind >= 0.70) & (kind < 0.75)] = 0
    obs_off = np.zeros(n_proj + 1, dtype=np.uint32)
    obs_off[1:] = np.cumsum(n_obs)
    total = int(obs_off[-1])
    obs_desc = random_descriptors(total, seed, 21)
    if nk and n_proj:
        target = (u64(seed, 0x6001, (n_proj,)) % np.uint64(nk)).astype(np.int64)
        ang = _unit(seed, 4, (n_proj,)) * 2 * np.pi
        rad = _unit(seed, 5, (n_proj,)) * 1.3 * radius
        near = kind < 0.70
        proj_xy[near] = kp_xy[target[near]] + np.stack([np.cos(ang), np.sin(ang)], axis=1)[near] * rad[near, None]
        owner = np.repeat(np.arange(n_proj), n_obs)
        strong = flip_masks(total, seed, 30, levels=(3, 4, 5))
        weak = flip_masks(total, seed, 31, levels=(2, 3))
        rel = (kind < 0.60)[owner]
        con = ((kind >= 0.60) & (kind < 0.70))[owner]
        obs_desc[rel] = kp_desc[target[owner[rel]]] ^ strong[rel]
        obs_desc[con] = kp_desc[target[owner[con]]] ^ weak[con]
    if lattice:
        kp_xy = np.floor(kp_xy)
        proj_xy = np.floor(proj_xy)
    return (np.ascontiguousarray(kp_desc), np.ascontiguousarray(kp_xy), np.ascontiguousarray(proj_xy),
            obs_off, np.ascontiguousarray(obs_desc))


def vocabulary(k: int, levels: int, seed: int, zero_weight_frac: float = 0.05, cluster_bits=(2, 3)):
    """A synthetic k-ary vocabulary tree of `levels` levels below the root (the reference's
    data/ORBvoc.cereal has k = 10, L = 6; that file is not in its checkout).  Children centroids
    are noisy copies of their parent's, so the tree clusters descriptor space like a real
    one; some leaves have weight 0 (bow_voc.h:114); nodes are stored parent-before-children.
    Returns the flattened arrays of include/vsmatch.h (vsm_voc_upload) as a dict."""
    n_nodes = sum(k ** l for l in range(levels + 1))
    node_desc = np.zeros((n_nodes, 4), dtype=np.uint64)
    child_off = np.zeros(n_nodes + 1, dtype=np.uint32)
    child_idx = []
    node_desc[0] = random_descriptors(1, seed, 40)[0]
    first, nxt = 0, 1
    for l in range(levels):
        width = k ** l
        kids = np.arange(nxt, nxt + width * k).reshape(width, k)
        masks = flip_masks(width * k, seed, 50 + l, levels=cluster_bits).reshape(width, k, 4)
        node_desc[kids] = node_desc[first:first + width, None, :] ^ masks
        child_idx.append(kids.reshape(-1))
        child_off[first + 1:first + width + 1] = child_off[first] + k * np.arange(1, width + 1)
        first, nxt = first + width, nxt + width * k
    child_off[first + 1:] = child_off[first]
    child_idx = np.concatenate(child_idx).astype(np.uint32) if child_idx else np.zeros(0, np.uint32)
    is_leaf = np.diff(child_off.astype(np.int64)) == 0
    word_id = np.zeros(n_nodes, dtype=np.uint32)
    word_id[is_leaf] = np.arange(int(is_leaf.sum()), dtype=np.uint32)  # createWords(), bow_voc.h:219-232
    weight = np.zeros(n_nodes, dtype=np.float64)
    u = _unit(seed, 60, (n_nodes,))
    weight[is_leaf] = np.where(u[is_leaf] < zero_weight_frac, 0.0, 0.5 + 10.0 * _unit(seed, 61, (n_nodes,))[is_leaf])
    return {"k": k, "L": levels, "node_desc": np.ascontiguousarray(node_desc), "child_off": child_off,
            "child_idx": child_idx, "node_id": np.arange(n_nodes, dtype=np.uint32), "word_id": word_id,
            "weight": weight, "parent": _parents(child_off, child_idx, n_nodes)}


def _parents(child_off, child_idx, n_nodes):
    parent = np.zeros(n_nodes, dtype=np.uint32)
    counts = np.diff(child_off.astype(np.int64))
    parent[child_idx] = np.repeat(np.arange(n_nodes, dtype=np.uint32), counts)
    return parent


def bow_vectors(n_img: int, n_words: int, words_per_image: int, seed: int, window: int = 8, share_frac: float = 0.5):
    """BoW vectors as BowVocabulary::transform returns them (bow_voc.h:94-131) -> (off uint32[n + 1],
    words uint32[total], values float64[total]): every image names `words_per_image` distinct words
    (fewer when draws collide) in no particular order, values positive and L1-normalised the way the
    reference does it (weights summed in order, each divided by the sum).  Image i takes
    `share_frac` of its words from a pool that moves with i (images closer than `window` overlap
    much, all images overlap a little through the frequent words), the rest uniformly."""
    off = np.zeros(n_img + 1, dtype=np.uint32)
    words, values = [], []
    m = int(words_per_image)
    n_shared = int(m * share_frac)
    pool = max(2 * m, 16)
    r = u64(seed, 0x7000, (n_img, m))
    wt = 0.5 + 10.0 * _unit(seed, 0x71, (n_img, m))
    for i in range(n_img):
        near = (np.uint64(i // max(window, 1) * (pool // 2)) + r[i, :n_shared] % np.uint64(pool)) % np.uint64(n_words)
        far = (r[i, n_shared:] >> np.uint64(20)) % np.uint64(n_words)
        w = np.concatenate([near, far]).astype(np.uint32)
        _, first = np.unique(w, return_index=True)
        keep = np.sort(first)  # order of first appearance
        w, x = w[keep], wt[i, keep]
        total = 0.0
        for v in x.tolist():
            total += v
        words.append(w)
        values.append(x / total)
        off[i + 1] = off[i] + len(w)
    return off, np.concatenate(words) if words else np.zeros(0, np.uint32), \
        np.concatenate(values) if values else np.zeros(0, np.float64)
