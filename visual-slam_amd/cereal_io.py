"""On-disk formats of the path's inputs and outputs.

The reference caches `KeypointsData` / `MatchData` maps with cereal
(/root/reference/include/visnav/serialization.h:176-205):

  * `corners.cereal`  = BinaryOutputArchive(feature_corners)   src/sfm.cpp:1203-1211
  * `matches.cereal`  = BinaryOutputArchive(feature_matches)   src/sfm.cpp:1262-1270, 1342-1350
  * `map.cereal`      = corners, matches, tracks, outlier tracks, cameras, landmarks
                        include/visnav/map_utils.h:58-116
  * JSON archives of the same types in the unit tests (test/src/test_ex3.cpp:148-185).

cereal itself is an un-vendored submodule, absent from /root/reference; this
module restates its published wire format (cereal 1.x, native little-endian
`BinaryOutputArchive`; `JSONOutputArchive` with `valueN` names for unnamed
values) for exactly the types on the path:

  size tag            uint64
  arithmetic          raw little-endian bytes
  vector<arithmetic>  size tag + contiguous elements
  vector<T>           size tag + elements
  pair<A, B>          A, B
  map-like<K, V>      size tag + (K, V) in the container's iteration order
  bitset<256>, binary uint8 type (3 = bits) + 32 bytes, bit i -> byte i/8, mask 0x80 >> i%8
  bitset<256>, JSON   {"type": 2, "data": "<256 chars>"} = std::bitset::to_string(), char 0 = bit 255
  Eigen fixed vector  its doubles (serialization.h:57-68)
  Sophus::SE3d        px py pz qx qy qz qw (serialization.h:155-164)
  FrameCamId          int64 frame_id, uint64 cam_id (serialization.h:202-205)

Descriptors come out as the matcher's packed layout: uint64[n, 4], bit k of the
bitset = bit k % 64 of word k / 64 (include/vsmatch.h).

Nothing here executes content of the files it reads.
"""
import json
import struct

import numpy as np

BITSET_TYPE_ULONG, BITSET_TYPE_ULLONG, BITSET_TYPE_STRING, BITSET_TYPE_BITS = 0, 1, 2, 3


# ----------------------------------------------------------------------------
# bitset<256> <-> packed uint64[4]

def pack_bits_msb_bytes(raw):
    """cereal binary bitset payload (n, 32) uint8 -> packed (n, 4) uint64."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1, 32)
    # byte c holds bits 8c .. 8c+7 with bit 8c at mask 0x80: reverse the bits of
    # every byte and the bytes are the little-endian image of the bitset.
    bits = np.unpackbits(raw, axis=1, bitorder="big")            # (n, 256): column i = bit i
    return np.packbits(bits, axis=1, bitorder="little").view("<u8").reshape(-1, 4)


def unpack_bits_msb_bytes(desc):
    """packed (n, 4) uint64 -> cereal binary bitset payload (n, 32) uint8."""
    desc = np.ascontiguousarray(desc, dtype="<u8").reshape(-1, 4)
    bits = np.unpackbits(desc.view(np.uint8).reshape(-1, 32), axis=1, bitorder="little")
    return np.packbits(bits, axis=1, bitorder="big")


def bitstring_to_packed(s):
    """std::bitset<256>::to_string() (char 0 = bit 255) -> uint64[4]."""
    if len(s) != 256 or set(s) - {"0", "1"}:
        raise ValueError("not a 256-character bit string")
    v = int(s, 2)
    return np.array([(v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)], dtype=np.uint64)


def packed_to_bitstring(d):
    v = 0
    for w in range(4):
        v |= int(d[w]) << (64 * w)
    return format(v, "0256b")


# ----------------------------------------------------------------------------
# binary archives

class BinaryReader:
    def __init__(self, data):
        self.b = memoryview(data)
        self.o = 0

    def _take(self, n):
        if self.o + n > len(self.b):
            raise ValueError(f"cereal binary archive truncated at byte {self.o} (+{n})")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def size(self):
        n = struct.unpack("<Q", self._take(8))[0]
        if n > len(self.b):  # every element takes at least one byte
            raise ValueError(f"implausible size tag {n} at byte {self.o - 8}")
        return n

    def i32(self):
        return struct.unpack("<i", self._take(4))[0]

    def i64(self):
        return struct.unpack("<q", self._take(8))[0]

    def u64(self):
        return struct.unpack("<Q", self._take(8))[0]

    def f64(self):
        return struct.unpack("<d", self._take(8))[0]

    def array(self, dtype, count):
        dt = np.dtype(dtype)
        return np.frombuffer(self._take(dt.itemsize * count), dtype=dt).copy()

    def at_end(self):
        return self.o == len(self.b)

    # --- the path's types
    def fcid(self):
        return (self.i64(), self.u64())

    def se3(self):
        return self.array("<f8", 7)

    def pairs_i32(self):
        n = self.size()
        return self.array("<i4", 2 * n).reshape(n, 2)

    def pairs_feature_track(self):
        """vector<pair<FeatureId, TrackId>>: int32 + int64 per element."""
        n = self.size()
        a = self.array(np.dtype([("f", "<i4"), ("t", "<i8")]), n)
        return np.stack([a["f"].astype(np.int64), a["t"]], axis=1) if n else np.zeros((0, 2), np.int64)

    def bitsets(self):
        n = self.size()
        rec = self.array(np.dtype([("type", "u1"), ("bits", "u1", 32)]), n)
        if n and not np.all(rec["type"] == BITSET_TYPE_BITS):
            raise ValueError("bitset<256> stored with a type other than `bits` in a binary archive")
        return pack_bits_msb_bytes(rec["bits"]) if n else np.zeros((0, 4), np.uint64)

    def keypoints(self):
        n = self.size()
        corners = self.array("<f8", 2 * n).reshape(n, 2)
        angles = self.array("<f8", self.size())
        return {"corners": corners, "corner_angles": angles, "corner_descriptors": self.bitsets()}

    def match_data(self):
        return {"T_i_j": self.se3(), "inliers": self.pairs_i32(), "matches": self.pairs_i32()}

    def landmark_match_data(self):
        return {"T_w_c": self.se3(), "inliers": self.pairs_feature_track(), "matches": self.pairs_feature_track()}

    def feature_track(self):
        n = self.size()
        a = self.array(np.dtype([("frame", "<i8"), ("cam", "<u8"), ("feat", "<i4")]), n)
        return [((int(r["frame"]), int(r["cam"])), int(r["feat"])) for r in a]

    def corners(self):
        return [(self.fcid(), self.keypoints()) for _ in range(self.size())]

    def matches(self):
        return [((self.fcid(), self.fcid()), self.match_data()) for _ in range(self.size())]

    def feature_tracks(self):
        return [(self.i64(), self.feature_track()) for _ in range(self.size())]

    def cameras(self):
        return [(self.fcid(), self.se3()) for _ in range(self.size())]

    def landmarks(self):
        out = []
        for _ in range(self.size()):
            tid = self.i64()
            p = self.array("<f8", 3)
            out.append((tid, {"p": p, "obs": self.feature_track(), "outlier_obs": self.feature_track()}))
        return out


class BinaryWriter:
    def __init__(self):
        self.parts = []

    def bytes(self):
        return b"".join(self.parts)

    def size(self, n):
        self.parts.append(struct.pack("<Q", n))

    def i64(self, v):
        self.parts.append(struct.pack("<q", v))

    def u64(self, v):
        self.parts.append(struct.pack("<Q", v))

    def array(self, a, dtype):
        self.parts.append(np.ascontiguousarray(a, dtype=dtype).tobytes())

    def fcid(self, f):
        self.i64(f[0])
        self.u64(f[1])

    def pairs_i32(self, p):
        p = np.asarray(p, dtype=np.int32).reshape(-1, 2)
        self.size(len(p))
        self.array(p, "<i4")

    def pairs_feature_track(self, p):
        p = np.asarray(p, dtype=np.int64).reshape(-1, 2)
        self.size(len(p))
        a = np.zeros(len(p), dtype=np.dtype([("f", "<i4"), ("t", "<i8")]))
        a["f"], a["t"] = p[:, 0], p[:, 1]
        self.parts.append(a.tobytes())

    def bitsets(self, desc):
        desc = np.asarray(desc, dtype=np.uint64).reshape(-1, 4)
        self.size(len(desc))
        rec = np.zeros(len(desc), dtype=np.dtype([("type", "u1"), ("bits", "u1", 32)]))
        rec["type"] = BITSET_TYPE_BITS
        if len(desc):
            rec["bits"] = unpack_bits_msb_bytes(desc)
        self.parts.append(rec.tobytes())

    def keypoints(self, kd):
        c = np.asarray(kd["corners"], dtype=np.float64).reshape(-1, 2)
        self.size(len(c))
        self.array(c, "<f8")
        a = np.asarray(kd["corner_angles"], dtype=np.float64)
        self.size(len(a))
        self.array(a, "<f8")
        self.bitsets(kd["corner_descriptors"])

    def match_data(self, md):
        self.array(md["T_i_j"], "<f8")
        self.pairs_i32(md["inliers"])
        self.pairs_i32(md["matches"])

    def landmark_match_data(self, md):
        self.array(md["T_w_c"], "<f8")
        self.pairs_feature_track(md["inliers"])
        self.pairs_feature_track(md["matches"])

    def feature_track(self, t):
        self.size(len(t))
        a = np.zeros(len(t), dtype=np.dtype([("frame", "<i8"), ("cam", "<u8"), ("feat", "<i4")]))
        for k, (f, feat) in enumerate(t):
            a[k] = (f[0], f[1], feat)
        self.parts.append(a.tobytes())

    def corners(self, items):
        self.size(len(items))
        for f, kd in items:
            self.fcid(f)
            self.keypoints(kd)

    def matches(self, items):
        self.size(len(items))
        for (f1, f2), md in items:
            self.fcid(f1)
            self.fcid(f2)
            self.match_data(md)

    def feature_tracks(self, items):
        self.size(len(items))
        for tid, t in items:
            self.i64(tid)
            self.feature_track(t)

    def cameras(self, items):
        self.size(len(items))
        for f, T in items:
            self.fcid(f)
            self.array(T, "<f8")

    def landmarks(self, items):
        self.size(len(items))
        for tid, lm in items:
            self.i64(tid)
            self.array(lm["p"], "<f8")
            self.feature_track(lm["obs"])
            self.feature_track(lm["outlier_obs"])


IDENTITY_SE3 = np.array([0, 0, 0, 0, 0, 0, 1], dtype=np.float64)  # px py pz qx qy qz qw


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def _finish(r, what, path):
    if not r.at_end():
        raise ValueError(f"{path}: {len(r.b) - r.o} trailing bytes after {what}")


def load_corners_binary(path):
    """`corners.cereal` (src/sfm.cpp:1203-1211): [((frame, cam), KeypointsData)] in file order."""
    r = BinaryReader(_read(path))
    out = r.corners()
    _finish(r, "Corners", path)
    return out


def save_corners_binary(path, items):
    w = BinaryWriter()
    w.corners(items)
    with open(path, "wb") as f:
        f.write(w.bytes())


def load_matches_binary(path):
    """`matches.cereal` (src/sfm.cpp:1262-1270): [(((f1, c1), (f2, c2)), MatchData)] in file order."""
    r = BinaryReader(_read(path))
    out = r.matches()
    _finish(r, "Matches", path)
    return out


def save_matches_binary(path, items):
    w = BinaryWriter()
    w.matches(items)
    with open(path, "wb") as f:
        f.write(w.bytes())


MAP_SECTIONS = ("corners", "matches", "feature_tracks", "outlier_tracks", "cameras", "landmarks")


def load_map_binary(path):
    """`map.cereal` (include/visnav/map_utils.h:88-116): the six archives in order."""
    r = BinaryReader(_read(path))
    out = {"corners": r.corners(), "matches": r.matches(), "feature_tracks": r.feature_tracks(),
           "outlier_tracks": r.feature_tracks(), "cameras": r.cameras(), "landmarks": r.landmarks()}
    _finish(r, "map", path)
    return out


def dump_map_binary(m):
    w = BinaryWriter()
    w.corners(m["corners"])
    w.matches(m["matches"])
    w.feature_tracks(m["feature_tracks"])
    w.feature_tracks(m["outlier_tracks"])
    w.cameras(m["cameras"])
    w.landmarks(m["landmarks"])
    return w.bytes()


def load_vocabulary_binary(path):
    """`ORBvoc.cereal` (BowVocabulary::save, bow_voc.h:133-144, 203-208): int32 m_k, int32 m_L,
    then m_nodes = size + per node (uint32 id, double weight, children = size + uint32s,
    uint32 parent, bitset<256> descriptor, uint32 word_id).  Word ids are re-assigned in node
    order over the leaves, as createWords() does after loading (bow_voc.h:219-232).  Returns the
    flattened arrays of vsm_voc_upload."""
    r = BinaryReader(_read(path))
    k, L = r.i32(), r.i32()
    n = r.size()
    node_id = np.zeros(n, np.uint32)
    weight = np.zeros(n, np.float64)
    parent = np.zeros(n, np.uint32)
    raw = np.zeros((n, 32), np.uint8)
    child_off = np.zeros(n + 1, np.uint32)
    kids = []
    for i in range(n):
        node_id[i] = r.array("<u4", 1)[0]
        weight[i] = r.f64()
        c = r.array("<u4", r.size())
        kids.append(c)
        child_off[i + 1] = child_off[i] + len(c)
        parent[i] = r.array("<u4", 1)[0]
        rec = r.array("u1", 33)
        if rec[0] != BITSET_TYPE_BITS:
            raise ValueError("bitset<256> stored with a type other than `bits` in a binary archive")
        raw[i] = rec[1:]
        r.array("<u4", 1)  # stored word_id, overwritten by createWords()
    _finish(r, "BowVocabulary", path)
    is_leaf = np.diff(child_off.astype(np.int64)) == 0
    word_id = np.zeros(n, np.uint32)
    word_id[is_leaf] = np.arange(int(is_leaf.sum()), dtype=np.uint32)
    return {"k": k, "L": L, "node_desc": pack_bits_msb_bytes(raw) if n else np.zeros((0, 4), np.uint64),
            "child_off": child_off, "child_idx": np.concatenate(kids).astype(np.uint32) if n and child_off[-1] else np.zeros(0, np.uint32),
            "node_id": node_id, "word_id": word_id, "weight": weight, "parent": parent}


def save_vocabulary_binary(path, voc):
    w = BinaryWriter()
    w.parts.append(struct.pack("<ii", int(voc["k"]), int(voc["L"])))
    n = len(voc["node_id"])
    w.size(n)
    payload = unpack_bits_msb_bytes(voc["node_desc"]) if n else np.zeros((0, 32), np.uint8)
    off = np.asarray(voc["child_off"], dtype=np.int64)
    for i in range(n):
        w.parts.append(struct.pack("<Id", int(voc["node_id"][i]), float(voc["weight"][i])))
        c = np.asarray(voc["child_idx"][off[i]:off[i + 1]], dtype="<u4")
        w.size(len(c))
        w.parts.append(c.tobytes())
        w.parts.append(struct.pack("<IB", int(voc["parent"][i]), BITSET_TYPE_BITS))
        w.parts.append(payload[i].tobytes())
        w.parts.append(struct.pack("<I", int(voc["word_id"][i])))
    with open(path, "wb") as f:
        f.write(w.bytes())


# ----------------------------------------------------------------------------
# JSON archives (cereal::JSONOutputArchive: unnamed values are "value0", "value1", ...)

def _json_bitsets(lst):
    out = np.zeros((len(lst), 4), dtype=np.uint64)
    for i, e in enumerate(lst):
        t = e["type"]
        if t == BITSET_TYPE_STRING:
            out[i] = bitstring_to_packed(e["data"])
        elif t in (BITSET_TYPE_ULONG, BITSET_TYPE_ULLONG):
            out[i, 0] = np.uint64(int(e["data"]))
        else:
            raise ValueError(f"bitset type {t} is not used by JSON archives")
    return out


def _json_pairs(lst):
    return np.array([[e["first"], e["second"]] for e in lst], dtype=np.int64).reshape(-1, 2)


def _json_se3(d):
    return np.array([d[k] for k in ("px", "py", "pz", "qx", "qy", "qz", "qw")], dtype=np.float64)


def keypoints_from_json(obj):
    """`archive(kd)` of a KeypointsData (test/ex3_test_data/kd0.json)."""
    v = obj["value0"] if "value0" in obj else obj
    corners = np.array([[c["value0"], c["value1"]] for c in v["m.corners"]], dtype=np.float64).reshape(-1, 2)
    return {"corners": corners, "corner_angles": np.array(v["m.corner_angles"], dtype=np.float64),
            "corner_descriptors": _json_bitsets(v["m.corner_descriptors"])}


def match_data_from_json(obj):
    """`archive(md)` of a MatchData / LandmarkMatchData (test/ex3_test_data/matches.json,
    test/ex5_test_data/matches.cereal -- JSON despite the name, test_ex5.cpp:167-171)."""
    v = obj["value0"] if "value0" in obj else obj
    key = "m.T_i_j" if "m.T_i_j" in v else "m.T_w_c"
    return {key[2:]: _json_se3(v[key]), "inliers": _json_pairs(v["m.inliers"]),
            "matches": _json_pairs(v["m.matches"])}


def load_keypoints_json(path):
    with open(path) as f:
        return keypoints_from_json(json.load(f))


def load_match_data_json(path):
    with open(path) as f:
        return match_data_from_json(json.load(f))


def load_projections_json(path):
    """test/ex5_test_data/projections.cereal: archive(projected_points); archive(track_ids)
    (test_ex5.cpp:117-122)."""
    with open(path) as f:
        o = json.load(f)
    pts = np.array([[c["value0"], c["value1"]] for c in o["value0"]], dtype=np.float64).reshape(-1, 2)
    return pts, np.array(o["value1"], dtype=np.int64)


def _fmt_double(x):
    # cereal writes doubles through RapidJSON's shortest round-trip printer, which
    # keeps a trailing ".0" on integral values -- Python's repr does the same.
    return repr(float(x))


def keypoints_to_json(kd, indent=4):
    v = {"m.corners": [{"value0": float(c[0]), "value1": float(c[1])} for c in np.asarray(kd["corners"]).reshape(-1, 2)],
         "m.corner_angles": [float(a) for a in kd["corner_angles"]],
         "m.corner_descriptors": [{"type": BITSET_TYPE_STRING, "data": packed_to_bitstring(d)}
                                  for d in np.asarray(kd["corner_descriptors"], dtype=np.uint64).reshape(-1, 4)]}
    return json.dumps({"value0": v}, indent=indent)


def match_data_to_json(md, indent=4):
    key = "T_i_j" if "T_i_j" in md else "T_w_c"
    T = md.get(key, IDENTITY_SE3)
    v = {"m." + key: dict(zip(("px", "py", "pz", "qx", "qy", "qz", "qw"), (float(x) for x in T))),
         "m.inliers": [{"first": int(a), "second": int(b)} for a, b in np.asarray(md["inliers"]).reshape(-1, 2)],
         "m.matches": [{"first": int(a), "second": int(b)} for a, b in np.asarray(md["matches"]).reshape(-1, 2)]}
    return json.dumps({"value0": v}, indent=indent)


# ----------------------------------------------------------------------------
# BoW database (bow_db.h:135-160: JSON archive of BowDBInverseIndex =
# unordered_map<WordId, vector<pair<FrameCamId, WordValue>>>, common_types.h:336-337) and the
# reference test's distance matrix (test_ex3.cpp:615-619: binary archive of a dynamic Eigen
# matrix of pair<FrameCamId, double>, serialization.h:57-90)

def bow_db_from_json(obj):
    """-> [(word, [((frame_id, cam_id), value), ...]), ...] in the archive's order."""
    return [(int(e["key"]), [((int(p["first"]["value0"]), int(p["first"]["value1"])), float(p["second"]))
                             for p in e["value"]]) for e in obj["value0"]]


def load_bow_db_json(path):
    with open(path) as f:
        return bow_db_from_json(json.load(f))


def bow_db_to_json(index, indent=4):
    """BowDatabase::save (bow_db.h:135-146)."""
    return json.dumps({"value0": [{"key": int(w), "value": [{"first": {"value0": int(fc[0]), "value1": int(fc[1])},
                                                            "second": float(v)} for fc, v in lst]}
                                  for w, lst in index]}, indent=indent)


def save_bow_db_json(path, index):
    with open(path, "w") as f:
        f.write(bow_db_to_json(index))


def bow_db_to_vectors(index):
    """The inverted index turned back into one BoW vector per image: {fcid: [(word, value), ...]}, the
    words of an image in the archive's order (the order inside the reference's BowVector is not
    recoverable from the index; it only decides the rounding of the L1 score)."""
    out = {}
    for w, lst in index:
        for fc, v in lst:
            out.setdefault(fc, []).append((w, v))
    return out


def load_bow_dist_binary(path):
    """-> (fcid int64[rows, cols, 2], dist float64[rows, cols])."""
    r = BinaryReader(_read(path))
    rows, cols = r.i32(), r.i32()
    if rows < 0 or cols < 0:
        raise ValueError("negative matrix size")
    rec = r.array(np.dtype([("f", "<i8"), ("c", "<u8"), ("d", "<f8")]), rows * cols).reshape(rows, cols)
    _finish(r, "bow distance matrix", path)
    return np.stack([rec["f"], rec["c"].astype(np.int64)], axis=-1), rec["d"].copy()


def bow_vectors_to_csr(vectors):
    """[[(word, value), ...], ...] -> (off uint32[n + 1], words uint32[total], values float64[total]):
    the `vsm_bowdb_upload` / `vsm_bowdb_query` layout (include/vsmatch.h)."""
    off = np.zeros(len(vectors) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(v) for v in vectors])
    words = np.array([w for v in vectors for w, _ in v], dtype=np.uint32)
    values = np.array([x for v in vectors for _, x in v], dtype=np.float64)
    return off, words, values


# ----------------------------------------------------------------------------
# flattening for the C ABI

def corners_to_database(items):
    """[(fcid, KeypointsData)] -> (fcids, desc uint64[total, 4], offsets uint32[n + 1]):
    the `vsm_db_upload` layout (include/vsmatch.h)."""
    fcids = [f for f, _ in items]
    sizes = [len(kd["corner_descriptors"]) for _, kd in items]
    offsets = np.zeros(len(items) + 1, dtype=np.uint32)
    offsets[1:] = np.cumsum(sizes)
    desc = (np.concatenate([np.asarray(kd["corner_descriptors"], dtype=np.uint64).reshape(-1, 4) for _, kd in items])
            if items else np.zeros((0, 4), np.uint64))
    return fcids, np.ascontiguousarray(desc), offsets
