"""ctypes bindings for the CPU checkers.  TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
leg.  The product package (visual-slam_amd/) never imports this.

  port()  -> oracle/libvsm_oracle.so   C restatement (oracle/vsm_oracle.c)
  ref()   -> oracle/_ref/libvsm_ref.so the real reference function compiled
             from /root/reference/include/visnav/keypoints.h:247-361, or None
             when it has not been built.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_f64p = C.POINTER(C.c_double)


def _as_desc(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim != 2 or a.shape[1] != 4:
        a = a.reshape(-1, 4)
    return a


class _Matcher:
    def __init__(self, lib, prefix, kind):
        self.kind = kind
        self._lib = lib
        self._match = getattr(lib, prefix + "_match")
        self._match.restype = C.c_size_t
        self._match.argtypes = [_u64p, C.c_size_t, _u64p, C.c_size_t, C.c_int,
                                C.c_double, _i32p, C.c_size_t]
        self._match_db = getattr(lib, prefix + "_match_db_pair")
        self._match_db.restype = C.c_size_t
        self._match_db.argtypes = [_u64p, _u32p, C.c_uint32, C.c_uint32, C.c_int,
                                   C.c_double, _i32p, C.c_size_t]
        self._nn2 = None
        if hasattr(lib, prefix + "_nn2"):
            self._nn2 = getattr(lib, prefix + "_nn2")
            self._nn2.restype = None
            self._nn2.argtypes = [_u64p, C.c_size_t, _u64p, C.c_size_t, _i32p, _i32p, _i32p]
        self._orb_a = self._orb_d = None
        if hasattr(lib, prefix + "_orb_angles"):
            _u8p = C.POINTER(C.c_uint8)
            self._orb_a = getattr(lib, prefix + "_orb_angles")
            self._orb_a.restype = None
            self._orb_a.argtypes = [_u8p, C.c_int, C.c_int, C.c_size_t, _f64p, C.c_size_t, C.c_int, _f64p]
            self._orb_d = getattr(lib, prefix + "_orb_descriptors")
            self._orb_d.restype = C.c_int
            self._orb_d.argtypes = [_u8p, C.c_int, C.c_int, C.c_size_t, _f64p, _f64p, C.c_size_t, _u64p]
        self._gftt = None
        if hasattr(lib, prefix + "_gftt"):
            _u8p = C.POINTER(C.c_uint8)
            self._gftt = getattr(lib, prefix + "_gftt")
            self._gftt.restype = C.c_long
            self._gftt.argtypes = [_u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_double, C.c_double, _f64p, C.c_size_t]
            self._detect = getattr(lib, prefix + "_detect_keypoints")
            self._detect.restype = C.c_long
            self._detect.argtypes = [_u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, _f64p, C.c_size_t]
            self._mineig = getattr(lib, prefix + "_gftt_mineig")
            self._mineig.restype = C.c_int
            self._mineig.argtypes = [_u8p, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_float)]
        self._bow = None
        if hasattr(lib, prefix + "_bow_words"):
            self._bow = getattr(lib, prefix + "_bow_words")
            self._bow.restype = C.c_int
            self._bow.argtypes = [_u64p, C.c_size_t, _u64p, _u32p, _u32p, _u32p, _u32p, _f64p, _u32p, _f64p]
        self._lm = None
        if hasattr(lib, prefix + "_match_landmarks"):
            self._lm = getattr(lib, prefix + "_match_landmarks")
            self._lm.restype = C.c_size_t
            self._lm.argtypes = [_u64p, _f64p, C.c_size_t, _f64p, C.c_size_t, _u32p, _u64p,
                                 C.c_double, C.c_int, C.c_double, _i32p, C.c_size_t]

    def match(self, d1, d2, threshold=70, dist_2_best=1.2) -> np.ndarray:
        """visnav::matchDescriptors -> int32[n_matches, 2] in reference order."""
        d1, d2 = _as_desc(d1), _as_desc(d2)
        cap = max(1, min(len(d1), len(d2)))
        out = np.empty((cap, 2), dtype=np.int32)
        n = self._match(d1.ctypes.data_as(_u64p), len(d1), d2.ctypes.data_as(_u64p),
                        len(d2), int(threshold), float(dist_2_best),
                        out.ctypes.data_as(_i32p), cap)
        if n == C.c_size_t(-1).value:
            raise MemoryError("oracle allocation failed")
        assert n <= cap, (n, cap)
        return out[:n].copy()

    def match_db_pair(self, db, offsets, i1, i2, threshold, dist_2_best, out) -> int:
        """One pair of a database; `out` is a caller-owned int32[cap, 2] buffer
        (so a thread pool can call this without allocating)."""
        return self._match_db(db.ctypes.data_as(_u64p), offsets.ctypes.data_as(_u32p),
                              int(i1), int(i2), int(threshold), float(dist_2_best),
                              out.ctypes.data_as(_i32p), out.shape[0])

    def nn2(self, q, t):
        """(best_idx, best_dist, second_dist) int32[nq] each (port only)."""
        if self._nn2 is None:
            raise NotImplementedError("nn2 is only restated in the C port")
        q, t = _as_desc(q), _as_desc(t)
        bi = np.empty(len(q), dtype=np.int32)
        bd = np.empty(len(q), dtype=np.int32)
        sd = np.empty(len(q), dtype=np.int32)
        self._nn2(q.ctypes.data_as(_u64p), len(q), t.ctypes.data_as(_u64p), len(t),
                  bi.ctypes.data_as(_i32p), bd.ctypes.data_as(_i32p),
                  sd.ctypes.data_as(_i32p))
        return bi, bd, sd


    def match_landmarks(self, kp_desc, kp_xy, proj_xy, obs_off, obs_desc, max_dist_2d=20.0,
                        threshold=70, dist_2_best=1.2) -> np.ndarray:
        """visnav::find_matches_landmarks -> int32[n, 2] (keypoint, projected point index);
        port only."""
        if self._lm is None:
            raise NotImplementedError("find_matches_landmarks is only restated in the C port")
        kp_desc, obs_desc = _as_desc(kp_desc), _as_desc(obs_desc)
        kp_xy = np.ascontiguousarray(kp_xy, dtype=np.float64).reshape(-1, 2)
        proj_xy = np.ascontiguousarray(proj_xy, dtype=np.float64).reshape(-1, 2)
        obs_off = np.ascontiguousarray(obs_off, dtype=np.uint32)
        assert len(kp_xy) == len(kp_desc) and len(obs_off) == len(proj_xy) + 1
        assert int(obs_off[-1]) == len(obs_desc)
        cap = max(1, len(kp_desc))
        out = np.empty((cap, 2), dtype=np.int32)
        n = self._lm(kp_desc.ctypes.data_as(_u64p), kp_xy.ctypes.data_as(_f64p), len(kp_desc),
                     proj_xy.ctypes.data_as(_f64p), len(proj_xy), obs_off.ctypes.data_as(_u32p),
                     obs_desc.ctypes.data_as(_u64p), float(max_dist_2d), int(threshold),
                     float(dist_2_best), out.ctypes.data_as(_i32p), cap)
        assert n <= cap
        return out[:n].copy()


    def bow_words(self, desc, voc):
        """BowVocabulary::transformFeatureToWord for every descriptor -> (word uint32[n],
        weight f64[n]); `voc` = dict of flattened tree arrays (visual-slam_amd/synth.vocabulary).
        Raises ValueError where the reference's loop would not terminate.  Port only."""
        desc = _as_desc(desc)
        n = len(desc)
        word = np.zeros(n, dtype=np.uint32)
        weight = np.zeros(n, dtype=np.float64)
        nd = _as_desc(voc["node_desc"])
        arr = {k: np.ascontiguousarray(voc[k], dtype=np.uint32) for k in ("child_off", "child_idx", "node_id", "word_id")}
        ci = arr["child_idx"] if len(arr["child_idx"]) else np.zeros(1, np.uint32)
        w = np.ascontiguousarray(voc["weight"], dtype=np.float64)
        rc = self._bow(desc.ctypes.data_as(_u64p), n, nd.ctypes.data_as(_u64p), arr["child_off"].ctypes.data_as(_u32p),
                       ci.ctypes.data_as(_u32p), arr["node_id"].ctypes.data_as(_u32p),
                       arr["word_id"].ctypes.data_as(_u32p), w.ctypes.data_as(_f64p), word.ctypes.data_as(_u32p),
                       weight.ctypes.data_as(_f64p))
        if rc != 0:
            raise ValueError("the reference's tree walk does not terminate on this input")
        return word, weight

    def gftt(self, img, max_corners=1500, quality=0.01, min_distance=8.0) -> np.ndarray:
        """cv::goodFeaturesToTrack -> float64[n, 2] (x, y) in detection order; port only."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        out = np.zeros((max(1, w * h), 2), dtype=np.float64)
        n = self._gftt(img.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, img.strides[0], int(max_corners),
                       float(quality), float(min_distance), out.ctypes.data_as(_f64p), len(out))
        assert 0 <= n <= len(out)
        return out[:n].copy()

    def detect_keypoints(self, img, num_features=1500) -> np.ndarray:
        """visnav::detectKeypoints -> kd.corners float64[n, 2]; port only."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        out = np.zeros((max(1, w * h), 2), dtype=np.float64)
        n = self._detect(img.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, img.strides[0], int(num_features),
                         out.ctypes.data_as(_f64p), len(out))
        assert 0 <= n <= len(out)
        return out[:n].copy()

    def gftt_mineig(self, img) -> np.ndarray:
        """The minimum-eigenvalue map of cv::cornerMinEigenVal(img, 3, 3) -> float32[h, w]."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        eig = np.zeros((h, w), dtype=np.float32)
        assert self._mineig(img.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, img.strides[0],
                            eig.ctypes.data_as(C.POINTER(C.c_float))) == 0
        return eig

    def orb_describe(self, img, corners_xy, rotate_features=True):
        """visnav::computeAngles + computeDescriptors -> (angles f64[n], desc uint64[n,4]);
        port only.  Raises ValueError for a corner inside the 19-pixel border."""
        if self._orb_a is None:
            raise NotImplementedError("ORB extraction is only restated in the C port")
        img = np.ascontiguousarray(img, dtype=np.uint8)
        assert img.ndim == 2
        xy = np.ascontiguousarray(corners_xy, dtype=np.float64).reshape(-1, 2)
        n = len(xy)
        ang = np.zeros(n, dtype=np.float64)
        desc = np.zeros((n, 4), dtype=np.uint64)
        u8 = C.POINTER(C.c_uint8)
        h, w = img.shape
        self._orb_a(img.ctypes.data_as(u8), w, h, img.strides[0], xy.ctypes.data_as(_f64p), n,
                    1 if rotate_features else 0, ang.ctypes.data_as(_f64p))
        rc = self._orb_d(img.ctypes.data_as(u8), w, h, img.strides[0], xy.ctypes.data_as(_f64p),
                         ang.ctypes.data_as(_f64p), n, desc.ctypes.data_as(_u64p))
        if rc != 0:
            raise ValueError("corner closer than EDGE_THRESHOLD to the border")
        return ang, desc


class BowDb:
    """visnav::BowDatabase (bow_db.h:45-168) as restated in oracle/vsm_oracle_bowdb.cpp; port only."""

    def __init__(self):
        lib = port()._lib
        self._lib = lib
        lib.vsm_oracle_bowdb_new.restype = C.c_void_p
        lib.vsm_oracle_bowdb_free.argtypes = [C.c_void_p]
        lib.vsm_oracle_bowdb_clear.argtypes = [C.c_void_p]
        lib.vsm_oracle_bowdb_insert.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, _u32p, _f64p, C.c_size_t]
        lib.vsm_oracle_bowdb_query.restype = C.c_size_t
        lib.vsm_oracle_bowdb_query.argtypes = [C.c_void_p, _u32p, _f64p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int64),
                                               C.POINTER(C.c_uint64), _f64p, C.c_size_t]
        self._h = C.c_void_p(lib.vsm_oracle_bowdb_new())

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.vsm_oracle_bowdb_free(self._h)
            self._h = None

    def clear(self):
        self._lib.vsm_oracle_bowdb_clear(self._h)

    @staticmethod
    def _vec(words, values):
        w = np.ascontiguousarray(words, dtype=np.uint32)
        v = np.ascontiguousarray(values, dtype=np.float64)
        assert w.shape == v.shape and w.ndim == 1
        return w, v

    def insert(self, fcid, words, values):
        w, v = self._vec(words, values)
        self._lib.vsm_oracle_bowdb_insert(self._h, int(fcid[0]), int(fcid[1]), w.ctypes.data_as(_u32p),
                                          v.ctypes.data_as(_f64p), len(w))

    def query(self, words, values, num_results):
        """-> (fcid int64[n, 2], score float64[n]) in the reference's result order."""
        w, v = self._vec(words, values)
        cap = 1024
        while True:
            fr, cam, sc = np.zeros(cap, np.int64), np.zeros(cap, np.uint64), np.zeros(cap, np.float64)
            n = self._lib.vsm_oracle_bowdb_query(self._h, w.ctypes.data_as(_u32p), v.ctypes.data_as(_f64p), len(w),
                                                 int(num_results), fr.ctypes.data_as(C.POINTER(C.c_int64)),
                                                 cam.ctypes.data_as(C.POINTER(C.c_uint64)), sc.ctypes.data_as(_f64p), cap)
            if n <= cap:
                return np.stack([fr[:n], cam[:n].astype(np.int64)], axis=1), sc[:n].copy()
            cap = n


def match_bow_pairs(fcids, vec_off, words, values, num_candidates):
    """The pair selection of match_bow (src/sfm.cpp:1366-1392) -> (pairs int32[n, 2], score float64[n]):
    (image, earlier image) in the reference's order, with the L1 score the query gave."""
    lib = port()._lib
    f = lib.vsm_oracle_match_bow_pairs
    f.restype = C.c_size_t
    f.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_uint64), _u32p, _u32p, _f64p, C.c_size_t, C.c_size_t, _i32p, _f64p, C.c_size_t]
    fr = np.ascontiguousarray([x[0] for x in fcids], dtype=np.int64)
    cam = np.ascontiguousarray([x[1] for x in fcids], dtype=np.uint64)
    off = np.ascontiguousarray(vec_off, dtype=np.uint32)
    w = np.ascontiguousarray(words, dtype=np.uint32)
    v = np.ascontiguousarray(values, dtype=np.float64)
    cap = max(1, len(fr) * int(num_candidates))
    pairs, sc = np.zeros((cap, 2), np.int32), np.zeros(cap, np.float64)
    n = f(fr.ctypes.data_as(C.POINTER(C.c_int64)), cam.ctypes.data_as(C.POINTER(C.c_uint64)), off.ctypes.data_as(_u32p),
          w.ctypes.data_as(_u32p), v.ctypes.data_as(_f64p), len(fr), int(num_candidates), pairs.ctypes.data_as(_i32p),
          sc.ctypes.data_as(_f64p), cap)
    assert n <= cap
    return pairs[:n].copy(), sc[:n].copy()


_cache = {}


def port() -> _Matcher:
    if "port" not in _cache:
        path = os.path.join(_HERE, "libvsm_oracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run `make -C oracle port`")
        _cache["port"] = _Matcher(C.CDLL(path), "vsm_oracle", "port")
    return _cache["port"]


def ref():
    if "ref" not in _cache:
        path = os.path.join(_HERE, "_ref", "libvsm_ref.so")
        _cache["ref"] = _Matcher(C.CDLL(path), "vsm_ref", "reference") \
            if os.path.exists(path) else None
    return _cache["ref"]
