"""BASELINE.json's full-size configurations inside the GPU suite, through the C ABI:
config 4 (1M x 1M, one direction: a slice against the oracle, size-independent properties on every
query), config 5 (one 32768-pair chunk of the 5000 x 2000 database: sampled pairs against the real
reference function where its build exists, else its port; the transposed chunk pair by pair), and
the on-disk formats on the device side (next row N3: a corners.cereal file in, a matches.cereal
file out, src/sfm.cpp:1203-1211, 1342-1350).  Bit-exact (integer and index work)."""
import importlib
import os

import numpy as np
import pytest

from oracle import binding
from tests import cases

pytestmark = pytest.mark.gpu
vsm = importlib.import_module("visual-slam_amd")
cio = importlib.import_module("visual-slam_amd.cereal_io")


def test_config4_one_million_by_one_million():
    """keypoints.h:256-290 at 2^20 x 2^20 (vsm_nn2_device, 32 train splits of 32768 rows merged on
    the device).  Every query: best index in range, best <= second, planted half found, the
    distance of the reported index recomputed on the host; 256 queries spread over the set against
    the oracle's full scan."""
    import torch
    n = 1 << 20
    q, t = cases.synth.planted_sets(n, seed=2027)
    dev = torch.device("cuda", 0)
    dq = torch.from_numpy(q.view(np.int64)).to(dev)
    dt = torch.from_numpy(t.view(np.int64)).to(dev)
    out = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    with vsm.Matcher(0) as m:
        m.nn2_device(dq.data_ptr(), n, dt.data_ptr(), n, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
        torch.cuda.synchronize()
    bi, bd, sd = (o.cpu().numpy() for o in out)
    assert bi.min() >= 0 and bi.max() < n
    assert np.all(bd <= sd) and bd.min() >= 0 and sd.max() <= 256
    assert (bd < 70).sum() > n // 3
    # the reported distance IS the distance to the reported row (vectorised popcount on the host)
    x = q ^ t[bi]
    pop = np.zeros(n, np.int64)
    for w in range(4):
        pop += np.unpackbits(x[:, w].copy().view(np.uint8).reshape(n, 8), axis=1).sum(axis=1, dtype=np.int64)
    assert np.array_equal(pop, bd)
    pick = np.linspace(0, n - 1, 256).astype(np.int64)
    wi, wd, ws = binding.port().nn2(q[pick], t)
    assert np.array_equal(bi[pick], wi) and np.array_equal(bd[pick], wd) and np.array_equal(sd[pick], ws)


def test_three_nn2_device_calls_in_flight():
    """vsm_nn2_device returns after enqueueing: three calls of different shapes back to back without
    a synchronisation in between (the job of each travels as a kernel argument, not through a host
    staging buffer the next call would overwrite), each against the oracle."""
    import torch
    dev = torch.device("cuda", 0)
    shapes = [(70000, 90000, 1), (300, 5000, 2), (9000, 40000, 3)]
    sets = [cases.synth.orb_like_pair(nq, nt, seed) for nq, nt, seed in shapes]
    d_in = [(torch.from_numpy(a.view(np.int64)).to(dev), torch.from_numpy(b.view(np.int64)).to(dev)) for a, b in sets]
    d_out = [[torch.full((len(a),), -9, dtype=torch.int32, device=dev) for _ in range(3)] for a, _ in sets]
    torch.cuda.synchronize()
    with vsm.Matcher(0) as m:
        for (da, db_), o, (a, b) in zip(d_in, d_out, sets):
            m.nn2_device(da.data_ptr(), len(a), db_.data_ptr(), len(b), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
        torch.cuda.synchronize()
    rng = np.random.default_rng(5)
    for (a, b), o in zip(sets, d_out):
        pick = rng.choice(len(a), size=min(len(a), 200), replace=False)
        want = binding.port().nn2(a[pick], b)
        for g, w in zip(o, want):
            assert np.array_equal(g.cpu().numpy()[pick], w), (len(a), len(b))
        assert int((o[0] >= 0).sum().cpu()) == len(a)  # every query written by ITS call


def test_config5_one_chunk_of_the_5000_x_2000_database():
    """The headline workload at full size: 5000 images x 2000 descriptors (320 MB) in HBM, one
    32768-pair chunk of the all-pairs list -- pairs of neighbouring images, which share landmarks,
    and of distant ones -- device resident, as bench.py runs it."""
    import torch
    db, off = cases.synth.sfm_database(5000, 2000, 2025, share_frac=0.4, window=8)
    dev = torch.device("cuda", 0)
    d_db = torch.from_numpy(db.view(np.int64)).to(dev)
    # 128 consecutive images x their 256 predecessors: 24576 pairs with overlap windows, + 8192 distant pairs
    i = np.repeat(np.arange(3000, 3096), 256)
    j = i - 1 - np.tile(np.arange(256), 96)
    far = np.stack([np.arange(8192) % 2000 + 2900, np.arange(8192) % 1777], axis=1)
    pairs = np.concatenate([np.stack([i, j], axis=1), far]).astype(np.uint32)
    P = len(pairs)
    assert P == 32768
    d_pairs = torch.from_numpy(pairs.view(np.int32).copy()).to(dev)
    d_rev = torch.from_numpy(pairs[:, ::-1].copy().view(np.int32)).to(dev)
    cap = P * 64
    d_off = torch.empty(P + 1, dtype=torch.int64, device=dev)
    d_out = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with vsm.Matcher(0) as m:
        m.db_attach(d_db.data_ptr(), off)
        total = m.match_pairs_device(d_pairs.data_ptr(), P, 70, 1.2, d_off.data_ptr(), d_out.data_ptr(), cap)
        offs, lists = d_off.cpu().numpy(), d_out[:total].cpu().numpy()
        total_r = m.match_pairs_device(d_rev.data_ptr(), P, 70, 1.2, d_off.data_ptr(), d_out.data_ptr(), cap)
        offs_r, lists_r = d_off.cpu().numpy(), d_out[:total_r].cpu().numpy()
    assert int(offs[-1]) == total == total_r and total > 100000
    cnt = np.diff(offs)
    assert np.array_equal(cnt, np.diff(offs_r))
    assert cnt[:24576].reshape(96, 256)[:, :7].min() > 50 and cnt[24576:].max() <= 2  # overlap only within the window
    # argument-swap property on every pair of the chunk: the transposed set
    pid = np.repeat(np.arange(P), cnt)
    a = np.stack([pid, lists[:, 0], lists[:, 1]], axis=1)
    b = np.stack([pid, lists_r[:, 1], lists_r[:, 0]], axis=1)
    b = b[np.lexsort((b[:, 1], b[:, 0]))]
    assert np.array_equal(a, b)
    impl = binding.ref() or binding.port()
    rng = np.random.default_rng(11)
    for k in np.concatenate([rng.choice(24576, 56, replace=False), 24576 + rng.choice(8192, 8, replace=False)]):
        p, r = pairs[k]
        want = impl.match(db[off[p]:off[p + 1]], db[off[r]:off[r + 1]], 70, 1.2)
        assert np.array_equal(lists[int(offs[k]):int(offs[k + 1])], want), (k, p, r)


def test_cereal_corners_file_in_matches_file_out(tmp_path):
    """The cached-pipeline round trip of `sfm` on the device side: corners.cereal --
    here the reference's own detector output for 20 EuRoC images, the first section of its binary
    map test/ex4_test_data/map_localize_camera.cereal, written by tests/golden/make_golden_n3.py --
    is read (serialization.h:176-205), flattened into the database layout, all image pairs of
    match_all (sfm.cpp:1285-1290) are matched on the GPU, the lists go into a matches.cereal archive
    (sfm.cpp:1342-1350) and come back from it: every list equals the oracle's."""
    items = cio.load_corners_binary(os.path.join(cases.GOLDEN, "ex4_corners.cereal"))
    assert len(items) == 20
    fcids, desc, off = cio.corners_to_database(items)
    assert len(desc) == 10037
    pairs = np.array([(i, j) for i in range(len(fcids)) for j in range(i) if fcids[i][0] != fcids[j][0]], np.uint32)
    assert len(pairs) == 180
    with vsm.Matcher(0) as m:
        m.db_upload(desc, off)
        offs, lists = m.match_pairs(pairs, 70, 1.2)
        where = {fc: k for k, fc in enumerate(fcids)}  # the file keeps the hash order of the reference's tbb map
        lr = [(where[(f, 0)], where[(f, 1)]) for f in range(10)]
        stereo = [m.match(desc[off[a]:off[a + 1]], desc[off[b]:off[b + 1]], 70, 1.2) for a, b in lr]
    out = [((fcids[i], fcids[j]), {"T_i_j": cio.IDENTITY_SE3, "inliers": np.zeros((0, 2), np.int32),
                                   "matches": lists[int(offs[k]):int(offs[k + 1])]})
           for k, (i, j) in enumerate(pairs.tolist())]
    path = tmp_path / "matches.cereal"
    cio.save_matches_binary(str(path), out)
    back = cio.load_matches_binary(str(path))
    assert len(back) == 180 and sum(len(md["matches"]) for _, md in back) > 3000
    o = binding.port()
    for k, ((key, md), (i, j)) in enumerate(zip(back, pairs.tolist())):
        assert key == (fcids[i], fcids[j])
        want = o.match(desc[off[i]:off[i + 1]], desc[off[j]:off[j + 1]], 70, 1.2)
        assert np.array_equal(md["matches"], want), (k, i, j)
    for f, (a, b) in enumerate(lr):  # match_stereo (sfm.cpp:1234-1244) on the same file
        assert np.array_equal(stereo[f], o.match(desc[off[a]:off[a + 1]], desc[off[b]:off[b + 1]], 70, 1.2)) and len(stereo[f]) > 50


def test_cpp_tool_corners_file_in_matches_file_out(tmp_path):
    """The same round trip from C++ (examples/vsm_sfm_match.cpp: cereal_lite.h + matchDescriptorsBatch,
    no cereal / Eigen / TBB): the binary built by build() reads the corners file, matches the pairs of
    match_all (sfm.cpp:1285-1321) resp. match_stereo (sfm.cpp:1234-1255) and writes matches.cereal;
    the Python reader gets every list back equal to the oracle's, keyed as the reference keys them."""
    import subprocess
    exe = os.path.join(os.path.dirname(cases.GOLDEN.rstrip("/")), "..", "examples", "vsm_sfm_match")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    src = os.path.join(cases.GOLDEN, "ex4_corners.cereal")
    fcids, desc, off = cio.corners_to_database(cio.load_corners_binary(src))
    where = {fc: k for k, fc in enumerate(fcids)}
    o = binding.port()
    for mode, n_pairs, thr, ratio in (("all", 180, 70, 1.2), ("stereo", 10, 80, 1.1)):
        path = str(tmp_path / (mode + ".cereal"))
        r = subprocess.run([exe, src, path, mode, str(thr), str(ratio)], capture_output=True, text=True, timeout=100)
        assert r.returncode == 0 and "vsm_sfm_match:" in r.stdout, r.stdout + r.stderr
        back = cio.load_matches_binary(path)
        assert len(back) == n_pairs
        for (a, b), md in back:
            i, j = where[a], where[b]
            assert (a[0] != b[0] and j < i) if mode == "all" else (a[0] == b[0] and a[1] == 0 and b[1] == 1)
            assert np.array_equal(md["matches"], o.match(desc[off[i]:off[i + 1]], desc[off[j]:off[j + 1]], thr, ratio)), (mode, a, b)
            assert len(md["inliers"]) == 0
        assert len({k for k, _ in back}) == n_pairs
    assert subprocess.run([exe, "/nonexistent", str(tmp_path / "x")], capture_output=True).returncode == 1
    # match_bow (sfm.cpp:1357-1423) through a vocabulary file (synthetic: the reference's ORBvoc.cereal is not in
    # its checkout): words by the oracle's descent, vectors, the oracle's BowDatabase loop, the oracle's matcher
    synth = importlib.import_module("visual-slam_amd.synth")
    voc = synth.vocabulary(8, 3, seed=5, zero_weight_frac=0.02)
    leaves = np.flatnonzero(np.diff(voc["child_off"].astype(np.int64)) == 0)
    rng = np.random.default_rng(8)  # descriptors near the leaves, so that the images share words
    kd = [(fc, {"corners": np.zeros((60, 2)), "corner_angles": np.zeros(60),
                "corner_descriptors": voc["node_desc"][leaves[rng.integers(0, len(leaves), 60)]]
                ^ synth.flip_masks(60, 9, k, levels=(4, 5))}) for k, fc in enumerate(fcids[:12])]
    cfile, vfile, mfile = (str(tmp_path / n) for n in ("c.cereal", "voc.cereal", "bow.cereal"))
    cio.save_corners_binary(cfile, kd)
    cio.save_vocabulary_binary(vfile, voc)
    r = subprocess.run([exe, cfile, mfile, "bow", vfile, "4", "90", "1.1"], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0 and "BoW candidate" in r.stdout, r.stdout + r.stderr
    d2 = np.concatenate([k["corner_descriptors"] for _, k in kd])
    off2 = (np.arange(13) * 60).astype(np.uint32)
    words, weights = o.bow_words(d2, voc)
    csr = vsm.matcher.bow_vectors_of_images(words, weights, off2)
    want_pairs, want_score = binding.match_bow_pairs(fcids[:12], *csr, 4)
    assert len(set(want_score.tolist())) == len(want_score) and len(want_pairs) > 10  # no equal scores: one right list
    back = cio.load_matches_binary(mfile)
    assert [(fcids.index(a), fcids.index(b)) for (a, b), _ in back] == [tuple(p) for p in want_pairs.tolist()]
    for ((a, b), md), (i, j) in zip(back, want_pairs.tolist()):
        assert np.array_equal(md["matches"], o.match(d2[off2[i]:off2[i + 1]], d2[off2[j]:off2[j + 1]], 90, 1.1))


def test_attach_again_after_rewriting_the_buffer():
    """vsm_db_attach reads the descriptors once (the group table of the consumer kernels): a caller
    that rewrites an attached buffer attaches again (include/vsmatch.h).  Two real-like databases
    through ONE device buffer, every list against the oracle both times."""
    import torch
    rng = np.random.default_rng(3)
    real = np.concatenate([np.load(os.path.join(cases.GOLDEN, "ex3_kd0.npy")), np.load(os.path.join(cases.GOLDEN, "ex3_kd1.npy"))])
    n_img, n_d = 48, 700
    off = (np.arange(n_img + 1) * n_d).astype(np.uint32)
    dbs = [np.ascontiguousarray(np.concatenate(
        [real[rng.choice(len(real), n_d, replace=False)] ^ cases.synth.flip_masks(n_d, 40 + r, i, levels=(5,)) for i in range(n_img)]))
        for r in range(2)]
    pairs = cases.synth.all_pairs(n_img)
    dev = torch.device("cuda", 0)
    buf = torch.empty((n_img * n_d, 4), dtype=torch.int64, device=dev)
    d_pairs = torch.from_numpy(pairs.view(np.int32).copy()).to(dev)
    d_off = torch.empty(len(pairs) + 1, dtype=torch.int64, device=dev)
    d_out = torch.empty((len(pairs) * n_d, 2), dtype=torch.int32, device=dev)
    o = binding.port()
    with vsm.Matcher(0) as m:
        m.set_variant(8)  # the pruned batched path, whatever the job count says
        for db in dbs:
            buf.copy_(torch.from_numpy(db.view(np.int64)))
            torch.cuda.synchronize()
            m.db_attach(buf.data_ptr(), off)
            total = m.match_pairs_device(d_pairs.data_ptr(), len(pairs), 70, 1.2, d_off.data_ptr(), d_out.data_ptr(), len(pairs) * n_d)
            offs, lists = d_off.cpu().numpy(), d_out[:total].cpu().numpy()
            assert total > 20000
            for k in range(0, len(pairs), 9):
                i, j = pairs[k]
                want = o.match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2)
                assert np.array_equal(lists[int(offs[k]):int(offs[k + 1])], want), (k, i, j)


def test_cpp_batch_of_200k_pairs_within_a_memory_bound(tmp_path):
    """matchDescriptorsBatch (the C++ drop-in for match_all, src/sfm.cpp:1276-1321) at config-5
    SHAPE -- 640 images x 2000 descriptors, all 204 480 pairs -- from the C++ binary.  A worst-case list buffer would
    be 204 480 x 2000 x 8 B = 3.3 GB of host memory (200 GB for the full 5000 images); the shim now sends the pair list
    through in chunks with buffers sized by the data.  Peak resident set under 2 GB; the lists of the single-context
    run and of the two-context run (vsm_match_pairs_multi) are identical (checksum over every list, in order), also
    with small chunks that force the capacity retry; every 997th pair against the oracle."""
    import subprocess
    root = os.path.dirname(cases.GOLDEN.rstrip("/"))
    exe = os.path.join(root, "cpp", "shim_batch_big_test")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    n_img, per, every = 640, 2000, 997
    db, off = cases.synth.sfm_database(n_img, per, seed=31, share_frac=0.4, window=8)
    path = tmp_path / "db.bin"
    db.tofile(path)
    lines = {}
    for name, extra in (("single", {}), ("two_contexts", {"VSM_DEVICES": "0,0"}),
                        ("small_chunks", {"VSM_BATCH_PAIRS": "50000"})):
        env = dict(os.environ)
        env.pop("VSM_DEVICES", None)
        env.pop("VSM_BATCH_PAIRS", None)
        env.update(extra)
        smp = tmp_path / (name + ".bin")
        r = subprocess.run([exe, str(path), str(n_img), str(per), str(every), str(smp)], capture_output=True, text=True,
                           timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        f = r.stdout.split()
        lines[name] = dict(zip(f[0::2], f[1::2]))
        assert int(lines[name]["pairs"]) == n_img * (n_img - 1) // 2 >= 200000
        assert int(lines[name]["peak_rss_mb"]) < 2048, lines[name]
    assert lines["single"]["checksum"] == lines["two_contexts"]["checksum"] == lines["small_chunks"]["checksum"]
    assert int(lines["single"]["matches"]) > 100000
    rec = np.fromfile(tmp_path / "single.bin", dtype=np.int32)
    port = binding.port()
    pos, seen = 0, 0
    while pos < len(rec):
        k, cnt = int(rec[pos]), int(rec[pos + 1])
        got = rec[pos + 2: pos + 2 + 2 * cnt].reshape(-1, 2)
        pos += 2 + 2 * cnt
        i = int((1 + np.sqrt(1 + 8 * k)) // 2)
        while i * (i - 1) // 2 > k:
            i -= 1
        while (i + 1) * i // 2 <= k:
            i += 1
        j = k - i * (i - 1) // 2
        want = port.match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2)
        assert np.array_equal(got, want), (k, i, j)
        seen += 1
    assert seen == (n_img * (n_img - 1) // 2 + every - 1) // every
