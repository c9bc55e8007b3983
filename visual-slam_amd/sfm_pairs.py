"""All-image-pair matching across the GPUs of one node.

Mirrors ``match_all`` of the reference (src/sfm.cpp:1276-1353): enumerate the
pairs (1285-1290), match every pair, collect the per-pair lists.  The reference
parallelises over pairs with ``tbb::parallel_for`` (1295); here the pair list
is cut block-cyclically over ranks (one process per GPU), the descriptor
database is replicated with ONE broadcast (RCCL over xGMI; ``backend="nccl"``
is RCCL on ROCm) and there is no communication while matching.  The result
lists are gathered to rank 0 afterwards (counts first, then the flat lists), device to device.

The per-rank matcher is injected (`match_pairs_fn`) so that the sharding and
gathering logic can be exercised on CPU with the gloo backend; the default is
the HIP path and it raises if the library or the GPU is missing.
"""
from __future__ import annotations

import numpy as np

from .synth import all_pairs  # noqa: F401  (sfm.cpp:1285-1290)

BLOCK = 4096  # pairs per block of the block-cyclic partition


def shard_indices(n_pairs: int, rank: int, world: int, block: int = BLOCK) -> np.ndarray:
    """Indices (into the canonical pair list) owned by `rank`: blocks of `block`
    consecutive pairs dealt round-robin.  Consecutive pairs share an image, so a
    block keeps that image's descriptors hot in the XCD L2s."""
    idx = np.arange(n_pairs, dtype=np.int64)
    return idx[(idx // block) % world == rank]


def broadcast_database(desc: np.ndarray | None, offsets: np.ndarray | None, device, group=None, src: int = 0):
    """Replicates (desc uint64[n,4], offsets uint32[n_img+1]) from `src` to every
    rank: one broadcast of sizes, one of offsets, one of the descriptors.
    Returns (desc_tensor on `device` as int64[n,4], offsets ndarray).  The tensor is complete
    when this returns (the device is synchronised): the matcher's context runs on a stream of
    its own, which does not wait for the collective's stream by itself (include/vsmatch.h)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == src:
        meta[0], meta[1] = desc.shape[0], offsets.shape[0]
    dist.broadcast(meta, src, group=group)
    n, n_off = int(meta[0]), int(meta[1])
    off_t = torch.zeros(n_off, dtype=torch.int64, device=device)
    d_t = torch.empty((n, 4), dtype=torch.int64, device=device)
    if rank == src:
        off_t.copy_(torch.from_numpy(offsets.astype(np.int64)))
        d_t.copy_(torch.from_numpy(np.ascontiguousarray(desc).view(np.int64)))
    dist.broadcast(off_t, src, group=group)
    dist.broadcast(d_t, src, group=group)
    off = off_t.cpu().numpy().astype(np.uint32)
    if d_t.is_cuda:
        torch.cuda.synchronize(d_t.device)
    return d_t, off


def gather_matches(local_counts, local_matches, n_pairs: int, group=None, dst: int = 0, block: int = BLOCK):
    """Gathers every rank's per-pair lists to `dst` in canonical pair order.

    `local_counts` (int tensor [n_local]): matches per pair of this rank's share, in the order
    of shard_indices(n_pairs, rank, world, block); `local_matches` (int32 tensor [total, 2]): its
    lists back to back.  Both live where the collectives of `group` want them (the GPU for RCCL,
    the CPU for gloo) and stay there: sizes are all-gathered, counts and lists are gathered padded
    to the largest share, and `dst` scatters every list to its place with vectorised index
    arithmetic (which pairs a rank owns follows from the partition, nothing but counts and lists
    travels).  Returns (offsets int64[n_pairs+1], matches int32[total,2]) tensors on dst, None
    elsewhere."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = local_matches.device
    local_counts = local_counts.to(device=device, dtype=torch.int64)
    local_matches = local_matches.to(dtype=torch.int32).reshape(-1, 2)
    sizes = torch.tensor([local_counts.numel(), local_matches.shape[0]], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = torch.stack(all_sizes).cpu().tolist()
    for r in range(world):
        want = len(shard_indices(n_pairs, r, world, block)) if n_pairs < (1 << 22) else None
        assert want is None or all_sizes[r][0] == want, "a rank's share does not follow the block-cyclic partition"
    max_p = max(1, max(s[0] for s in all_sizes))
    max_m = max(1, max(s[1] for s in all_sizes))
    head = torch.zeros(max_p, dtype=torch.int64, device=device)
    head[: local_counts.numel()] = local_counts
    body = torch.zeros((max_m, 2), dtype=torch.int32, device=device)
    body[: local_matches.shape[0]] = local_matches
    heads = [torch.zeros_like(head) for _ in range(world)] if rank == dst else None
    bodies = [torch.zeros_like(body) for _ in range(world)] if rank == dst else None
    dist.gather(head, heads, dst=dst, group=group)
    dist.gather(body, bodies, dst=dst, group=group)
    if rank != dst:
        return None
    idx = [torch.from_numpy(shard_indices(n_pairs, r, world, block)).to(device) for r in range(world)]
    cnt = torch.zeros(n_pairs, dtype=torch.int64, device=device)
    for r in range(world):
        cnt[idx[r]] = heads[r][: all_sizes[r][0]]
    offsets = torch.zeros(n_pairs + 1, dtype=torch.int64, device=device)
    offsets[1:] = torch.cumsum(cnt, 0)
    out = torch.empty((int(offsets[-1]), 2), dtype=torch.int32, device=device)
    for r in range(world):
        p, m = all_sizes[r]
        if m == 0:
            continue
        c = heads[r][:p]
        local_start = torch.cumsum(c, 0) - c
        # match k of rank r, in local pair q: global row = offsets[idx[q]] + (k - local_start[q])
        shift = torch.repeat_interleave(offsets[idx[r]] - local_start, c)
        out[shift + torch.arange(m, device=device)] = bodies[r][:m]
    return offsets, out


def _hip_match_pairs(device_index: int):
    from .matcher import Matcher
    m = Matcher(device_index)

    def run(desc, offsets, pairs, threshold, dist_2_best):
        m.db_upload(desc, offsets)
        return m.match_pairs(pairs, threshold, dist_2_best)

    return run


def match_all_sharded(desc, offsets, pairs, threshold: int = 70, dist_2_best: float = 1.2, group=None,
                      device="cpu", match_pairs_fn=None, block: int = BLOCK):
    """Every rank matches its block-cyclic share of `pairs`; rank 0 returns
    (offsets, matches) for the whole list in canonical order, other ranks None.
    `desc`/`offsets`/`pairs` must be identical on every rank (replicated
    database, see broadcast_database).

    `match_pairs_fn(desc, offsets, pairs, threshold, dist_2_best) -> (offsets, matches)`
    defaults to the HIP path on this rank's GPU."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if match_pairs_fn is None:
        import torch
        match_pairs_fn = _hip_match_pairs(torch.cuda.current_device())
    import torch
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    mine = shard_indices(len(pairs), rank, world, block)
    loc_off, loc_matches = match_pairs_fn(desc, offsets, pairs[mine], threshold, dist_2_best)
    counts = torch.from_numpy(np.diff(np.asarray(loc_off).astype(np.int64))).to(device)
    lists = torch.from_numpy(np.ascontiguousarray(loc_matches, dtype=np.int32).reshape(-1, 2)).to(device)
    res = gather_matches(counts, lists, len(pairs), group, block=block)
    if res is None:
        return None
    return res[0].cpu().numpy().astype(np.uint64), res[1].cpu().numpy()


# ---- 1M x 1M style one-direction matching (BASELINE.json config 4) -------------------------
def shard_range(n: int, rank: int, world: int):
    """Contiguous query block of `rank`: [lo, hi)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def nn2_sharded(query, train, group=None, device="cpu", nn2_fn=None):
    """keypoints.h:256-290 for a huge query set: queries are sharded over ranks in
    contiguous blocks, the train set is replicated (every rank passes the same
    `train`), no merge is needed; one all-gather returns the full
    (best_idx, best_dist, second_dist) int32[nq] on every rank (the cross-check
    variant needs the other direction's best_idx everywhere).

    `nn2_fn(query_block, train) -> (best_idx, best_dist, second_dist)` defaults to
    the HIP path on this rank's GPU."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if nn2_fn is None:
        from .matcher import Matcher
        m = Matcher(torch.cuda.current_device())
        nn2_fn = m.nn2
    nq = len(query)
    lo, hi = shard_range(nq, rank, world)
    bi, bd, sd = nn2_fn(query[lo:hi], train)
    per = (nq + world - 1) // world
    mine = torch.zeros((3, max(per, 1)), dtype=torch.int32, device=device)
    if hi > lo:
        mine[0, : hi - lo] = torch.from_numpy(np.ascontiguousarray(bi, dtype=np.int32))
        mine[1, : hi - lo] = torch.from_numpy(np.ascontiguousarray(bd, dtype=np.int32))
        mine[2, : hi - lo] = torch.from_numpy(np.ascontiguousarray(sd, dtype=np.int32))
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    full = torch.cat(parts, dim=1)[:, :nq].cpu().numpy()
    return full[0], full[1], full[2]


def match_bow(matcher, desc, offsets, frame_ids, num_bow_candidates: int = 25, threshold: int = 70,
              dist_2_best: float = 1.2, max_features_per_call: int = 1 << 23):
    """`match_bow` of the reference (src/sfm.cpp:1357-1423) on one GPU, for a descriptor database in
    the `vsm_db_upload` layout and a vocabulary already uploaded (`matcher.voc_upload`): every
    descriptor's word (bow_voc.h:57-92, GPU), every image's BoW vector (bow_voc.h:107-130, host: a few
    hundred ordered double additions per image), the candidate pairs (query-then-insert over a BoW
    database, sfm.cpp:1373-1392, GPU), and matchDescriptors on those pairs (sfm.cpp:1409-1411, GPU).
    -> (pairs uint32[n, 2], out_offsets int64[n + 1], matches int32[total, 2]).
    The caller's next step (findInliersRansac, sfm.cpp:1413-1418) is outside this path."""
    from .matcher import as_descriptors, bow_vectors_of_images
    desc = as_descriptors(desc)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    words = np.zeros(len(desc), dtype=np.uint32)
    weights = np.zeros(len(desc), dtype=np.float64)
    for a in range(0, len(desc), max_features_per_call):
        words[a:a + max_features_per_call], weights[a:a + max_features_per_call] = \
            matcher.bow_words(desc[a:a + max_features_per_call])
    matcher.bowdb_upload(bow_vectors_of_images(words, weights, offsets))
    pairs = matcher.bow_select_pairs(frame_ids, num_bow_candidates)
    matcher.db_upload(desc, offsets)
    out_offsets, matches = matcher.match_pairs(pairs, threshold, dist_2_best)
    return pairs, out_offsets, matches
