"""Host-side mirror of the reference interface over the C ABI (include/vsmatch.h).

``match_descriptors`` has the argument meaning of ``visnav::matchDescriptors``
(/root/reference/include/visnav/keypoints.h:247-250); ``Matcher.match_pairs``
is the batched form of the ``match_all`` loop (src/sfm.cpp:1295-1321).

There is no CPU path here: if ``libvsmatch.so`` is missing or no MI355X is
present, every call raises (``VsmError``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvsmatch.so")

VSM_OK, VSM_ERR_INVALID, VSM_ERR_HIP, VSM_ERR_CAPACITY, VSM_ERR_NOMEM = range(5)
VSM_MAX_SET = 1 << 23

# every symbol include/vsmatch.h declares
SYMBOLS = (
    "vsm_version", "vsm_last_error", "vsm_create", "vsm_destroy", "vsm_stream",
    "vsm_match_pair", "vsm_match_pair_noalloc", "vsm_nn2", "vsm_nn2_device", "vsm_db_upload", "vsm_db_attach",
    "vsm_match_pairs", "vsm_match_pairs_device", "vsm_db_upload_multi", "vsm_match_pairs_multi", "vsm_last_scan_stats", "vsm_last_scan_stats_kind", "vsm_last_scan_kernel",
    "vsm_set_profiling",
    "vsm_set_variant", "vsm_accept_limit",
    "vsm_match_landmarks",
    "vsm_orb_describe", "vsm_orb_angles", "vsm_orb_descriptors",
    "vsm_detect_keypoints", "vsm_good_features_to_track", "vsm_detect_and_describe", "vsm_detect_and_describe_batch",
    "vsm_voc_upload", "vsm_bow_words",
    "vsm_bowdb_upload", "vsm_bowdb_query", "vsm_bow_select_pairs",
)


class VsmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvsmatch error {code}: {msg}")
        self.code = code


_lib = None
_lib_lock = threading.Lock()
_vp = C.c_void_p
_sz = C.c_size_t


def _share_the_hip_runtime_with_torch():
    """PyTorch's ROCm wheels ship their own libamdhip64.so.7 (and HSA runtime); libvsmatch.so links
    the system's /opt/rocm one of the same SONAME.  A process can hold only one of them: whichever is
    loaded first serves both, and PyTorch on the system's newer runtime fails ("No HIP GPUs are
    available").  So when PyTorch is installed -- whether or not it has been imported yet -- its
    runtime is loaded first and libvsmatch.so binds to it.  Without PyTorch: the system's."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Loads libvsmatch.so (built by __graft_entry__.build()); raises if absent."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise VsmError(VSM_ERR_HIP, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
            _share_the_hip_runtime_with_torch()
            L = C.CDLL(LIB_PATH)
            L.vsm_version.restype = C.c_int
            L.vsm_last_error.restype = C.c_char_p
            L.vsm_create.argtypes = [C.c_int, C.POINTER(_vp)]
            L.vsm_destroy.argtypes = [_vp]
            L.vsm_destroy.restype = None
            L.vsm_stream.argtypes = [_vp]
            L.vsm_stream.restype = _vp
            L.vsm_match_pair.argtypes = [_vp, _vp, _sz, _vp, _sz, C.c_int, C.c_double, _vp, _sz, C.POINTER(_sz)]
            L.vsm_match_pair_noalloc.argtypes = [_vp, _vp, _sz, _vp, _sz, C.c_int, C.c_double, _vp, _sz, _vp]
            L.vsm_nn2.argtypes = [_vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp]
            L.vsm_nn2_device.argtypes = [_vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp]
            L.vsm_db_upload.argtypes = [_vp, _vp, _vp, _sz]
            L.vsm_db_attach.argtypes = [_vp, _vp, _vp, _sz]
            L.vsm_match_pairs.argtypes = [_vp, _vp, _sz, C.c_int, C.c_double, _vp, _vp, _sz, C.POINTER(_sz)]
            L.vsm_match_pairs_device.argtypes = [_vp, _vp, _sz, C.c_int, C.c_double, _vp, _vp, _sz, C.POINTER(_sz)]
            L.vsm_db_upload_multi.argtypes = [C.POINTER(_vp), _sz, _vp, _vp, _sz]
            L.vsm_match_pairs_multi.argtypes = [C.POINTER(_vp), _sz, _vp, _sz, C.c_int, C.c_double, _vp, _vp, _sz,
                                                C.POINTER(_sz)]
            L.vsm_last_scan_stats.argtypes = [_vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
            L.vsm_last_scan_kernel.argtypes = [_vp, C.c_int]
            L.vsm_last_scan_stats_kind.argtypes = [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
            L.vsm_set_profiling.argtypes = [_vp, C.c_int]
            L.vsm_set_variant.argtypes = [_vp, C.c_int]
            L.vsm_accept_limit.argtypes = [C.c_int, C.c_double]
            L.vsm_accept_limit.restype = C.c_int
            L.vsm_orb_describe.argtypes = [_vp, _vp, C.c_int, C.c_int, _sz, _vp, _sz, C.c_int, _vp, _vp]
            L.vsm_detect_keypoints.argtypes = [_vp, _vp, C.c_int, C.c_int, _sz, C.c_int, _vp, _sz, C.POINTER(_sz)]
            L.vsm_good_features_to_track.argtypes = [_vp, _vp, C.c_int, C.c_int, _sz, C.c_int, C.c_double, C.c_double,
                                                     _vp, _sz, C.POINTER(_sz)]
            L.vsm_detect_and_describe_batch.argtypes = [_vp, _vp, _sz, C.c_int, C.c_int, _sz, C.c_int, C.c_int, _vp, _vp, _vp,
                                                        _vp, _sz, C.c_int]
            L.vsm_detect_and_describe.argtypes = [_vp, _vp, C.c_int, C.c_int, _sz, C.c_int, C.c_int, _vp, _vp, _vp,
                                                  _sz, C.POINTER(_sz)]
            L.vsm_voc_upload.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]
            L.vsm_bow_words.argtypes = [_vp, _vp, _sz, _vp, _vp]
            L.vsm_bowdb_upload.argtypes = [_vp, _vp, _vp, _vp, _sz]
            L.vsm_bowdb_query.argtypes = [_vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp]
            L.vsm_bow_select_pairs.argtypes = [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]
            L.vsm_orb_angles.argtypes = [_vp, _vp, C.c_int, C.c_int, _sz, _vp, _sz, C.c_int, _vp]
            L.vsm_orb_descriptors.argtypes = [_vp, _vp, C.c_int, C.c_int, _sz, _vp, _vp, _sz, _vp]
            L.vsm_match_landmarks.argtypes = [_vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp, C.c_double, C.c_int,
                                              C.c_double, _vp, _sz, C.POINTER(_sz)]
            _lib = L
    return _lib


def _check(rc, allow=()):
    if rc != VSM_OK and rc not in allow:
        raise VsmError(rc, lib().vsm_last_error().decode("utf-8", "replace"))
    return rc


def as_descriptors(a) -> np.ndarray:
    """uint64[n, 4] C-contiguous view/copy of packed descriptors."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim == 1:
        if a.size % 4:
            raise ValueError("descriptor array length must be a multiple of 4 uint64 words")
        a = a.reshape(-1, 4)
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError(f"descriptors must be uint64[n, 4], got {a.shape}")
    return a


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a.size else C.c_void_p(0)


class Matcher:
    """One vsm_ctx = one GPU + one HIP stream + library-owned device scratch."""

    def __init__(self, device: int = 0):
        self._h = _vp()
        _check(lib().vsm_create(int(device), C.byref(self._h)))
        self.device = int(device)
        self._offsets = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().vsm_destroy(self._h)
            self._h = _vp()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def stream(self) -> int:
        return lib().vsm_stream(self._h) or 0

    # -- keypoints.h:247-250 ------------------------------------------------
    def match(self, d1, d2, threshold: int = 70, dist_2_best: float = 1.2) -> np.ndarray:
        """visnav::matchDescriptors on one pair -> int32[n_matches, 2]."""
        d1, d2 = as_descriptors(d1), as_descriptors(d2)
        cap = max(1, min(len(d1), len(d2)))
        out = np.empty((cap, 2), dtype=np.int32)
        n = _sz(0)
        _check(lib().vsm_match_pair(self._h, _ptr(d1), len(d1), _ptr(d2), len(d2), int(threshold),
                                    float(dist_2_best), _ptr(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def match_pair_noalloc(self, d_d1: int, n1: int, d_d2: int, n2: int, threshold: int, dist_2_best: float,
                           d_out_pairs: int, cap: int, d_out_count: int):
        """vsm_match_pair_noalloc: raw device pointers in and out, only enqueues (synchronise
        `stream` before reading), allocates nothing; at most 8192 descriptors per side."""
        _check(lib().vsm_match_pair_noalloc(self._h, _vp(d_d1), n1, _vp(d_d2), n2, int(threshold), float(dist_2_best),
                                            _vp(d_out_pairs), int(cap), _vp(d_out_count)))

    # -- keypoints.h:256-290 --------------------------------------------------
    def nn2(self, query, train):
        """(best_idx, best_dist, second_dist), int32[nq] each."""
        q, t = as_descriptors(query), as_descriptors(train)
        bi = np.empty(len(q), dtype=np.int32)
        bd = np.empty(len(q), dtype=np.int32)
        sd = np.empty(len(q), dtype=np.int32)
        _check(lib().vsm_nn2(self._h, _ptr(q), len(q), _ptr(t), len(t), _ptr(bi), _ptr(bd), _ptr(sd)))
        return bi, bd, sd

    def nn2_device(self, d_query: int, nq: int, d_train: int, nt: int, d_best_idx: int,
                   d_best_dist: int, d_second_dist: int):
        """Raw device pointers (e.g. torch tensors' data_ptr()); asynchronous."""
        _check(lib().vsm_nn2_device(self._h, _vp(d_query), nq, _vp(d_train), nt, _vp(d_best_idx),
                                    _vp(d_best_dist), _vp(d_second_dist)))

    # -- src/sfm.cpp:1191-1214 feature_corners, flattened ---------------------
    def db_upload(self, desc, offsets):
        desc = as_descriptors(desc)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        if offsets.ndim != 1 or len(offsets) < 1 or int(offsets[-1]) != len(desc):
            raise ValueError("offsets must be uint32[n_img+1] ending at len(desc)")
        _check(lib().vsm_db_upload(self._h, _ptr(desc), _ptr(offsets), len(offsets) - 1))
        self._offsets = offsets

    def db_attach(self, d_desc: int, offsets):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        _check(lib().vsm_db_attach(self._h, _vp(d_desc), _ptr(offsets), len(offsets) - 1))
        self._offsets = offsets

    # -- src/sfm.cpp:1295-1321 ------------------------------------------------
    def match_pairs(self, pairs, threshold: int = 70, dist_2_best: float = 1.2, cap: int | None = None):
        """All pairs at once -> (offsets uint64[n+1], matches int32[total, 2]);
        pair k's list is matches[offsets[k]:offsets[k+1]]."""
        if self._offsets is None:
            raise VsmError(VSM_ERR_INVALID, "no database uploaded")
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        n = len(pairs)
        if cap is None:
            sizes = np.diff(self._offsets.astype(np.int64))
            ok = (pairs < len(sizes)).all(axis=1) if n else np.zeros(0, bool)
            cap = int(np.minimum(sizes[pairs[ok, 0]], sizes[pairs[ok, 1]]).sum()) if n else 0
        cap = max(int(cap), 1)
        offs = np.empty(n + 1, dtype=np.uint64)
        out = np.empty((cap, 2), dtype=np.int32)
        total = _sz(0)
        _check(lib().vsm_match_pairs(self._h, _ptr(pairs), n, int(threshold), float(dist_2_best),
                                     _ptr(offs), _ptr(out), cap, C.byref(total)))
        return offs, out[: total.value].copy()

    def match_pairs_device(self, d_pairs: int, n_pairs: int, threshold: int, dist_2_best: float,
                           d_out_offsets: int, d_out_pairs: int, cap: int) -> int:
        """Everything device resident; returns the total number of matches.
        Raises VsmError(VSM_ERR_CAPACITY) if it exceeds `cap`."""
        total = _sz(0)
        _check(lib().vsm_match_pairs_device(self._h, _vp(d_pairs), n_pairs, int(threshold),
                                            float(dist_2_best), _vp(d_out_offsets), _vp(d_out_pairs),
                                            int(cap), C.byref(total)))
        return total.value

    # -- vo_utils.h:95-187 ------------------------------------------------------
    def match_landmarks(self, kp_desc, kp_xy, proj_xy, obs_off, obs_desc, match_max_dist_2d: float,
                        threshold: int = 70, dist_2_best: float = 1.2, track_ids=None) -> np.ndarray:
        """visnav::find_matches_landmarks on flattened containers (see include/vsmatch.h).
        Returns int32[n, 2] (key_point_index, projected_point_index), or with `track_ids`
        (projected_track_ids) int64[n, 2] (featureId, trackId) = md.matches."""
        kp_desc, obs_desc = as_descriptors(kp_desc), as_descriptors(obs_desc)
        kp_xy = np.ascontiguousarray(kp_xy, dtype=np.float64).reshape(-1, 2)
        proj_xy = np.ascontiguousarray(proj_xy, dtype=np.float64).reshape(-1, 2)
        obs_off = np.ascontiguousarray(obs_off, dtype=np.uint32)
        if len(kp_xy) != len(kp_desc):
            raise ValueError("kp_xy and kp_desc differ in length")
        if obs_off.ndim != 1 or len(obs_off) != len(proj_xy) + 1 or int(obs_off[-1]) != len(obs_desc):
            raise ValueError("obs_off must be uint32[np + 1] ending at len(obs_desc)")
        cap = max(1, len(kp_desc))
        out = np.empty((cap, 2), dtype=np.int32)
        n = _sz(0)
        _check(lib().vsm_match_landmarks(self._h, _ptr(kp_desc), _ptr(kp_xy), len(kp_desc), _ptr(proj_xy),
                                         len(proj_xy), _ptr(obs_off), _ptr(obs_desc), float(match_max_dist_2d),
                                         int(threshold), float(dist_2_best), _ptr(out), cap, C.byref(n)))
        got = out[: n.value].copy()
        if track_ids is None:
            return got
        tid = np.asarray(track_ids, dtype=np.int64)
        return np.stack([got[:, 0].astype(np.int64), tid[got[:, 1]]], axis=1)

    # -- keypoints.h:151-237 ------------------------------------------------------
    def orb_describe(self, img, corners_xy, rotate_features: bool = True, descriptors: bool = True):
        """visnav::computeAngles + computeDescriptors on an 8-bit image (2-D uint8 array, rows may
        be strided) -> (corner_angles f64[n], corner_descriptors uint64[n, 4] or None)."""
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1:
            img = np.ascontiguousarray(img, dtype=np.uint8)
            if img.ndim != 2:
                raise ValueError("image must be 2-D uint8")
        xy = np.ascontiguousarray(corners_xy, dtype=np.float64).reshape(-1, 2)
        n = len(xy)
        ang = np.zeros(n, dtype=np.float64)
        desc = np.zeros((n, 4), dtype=np.uint64) if descriptors else None
        h, w = img.shape
        _check(lib().vsm_orb_describe(self._h, C.c_void_p(img.ctypes.data), w, h, img.strides[0], _ptr(xy), n,
                                      1 if rotate_features else 0, _ptr(ang),
                                      _ptr(desc) if descriptors else C.c_void_p(0)))
        return ang, desc

    # -- keypoints.h:133-149, 239-245 ----------------------------------------------
    @staticmethod
    def _image(img):
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1:
            img = np.ascontiguousarray(img, dtype=np.uint8)
            if img.ndim != 2:
                raise ValueError("image must be 2-D uint8")
        return img

    def good_features_to_track(self, img, max_corners: int, quality_level: float = 0.01,
                               min_distance: float = 8.0) -> np.ndarray:
        """cv::goodFeaturesToTrack -> float64[n, 2] (x, y), strongest first."""
        img = self._image(img)
        h, w = img.shape
        cap = max_corners if max_corners > 0 else w * h
        n = _sz(0)
        while True:
            out = np.zeros((max(cap, 1), 2), dtype=np.float64)
            rc = _check(lib().vsm_good_features_to_track(self._h, C.c_void_p(img.ctypes.data), w, h, img.strides[0],
                                                         int(max_corners), float(quality_level), float(min_distance),
                                                         _ptr(out), cap, C.byref(n)), allow=(VSM_ERR_CAPACITY,))
            if rc == VSM_OK:
                return out[: n.value].copy()
            cap = n.value

    def detect_keypoints(self, img, num_features: int) -> np.ndarray:
        """visnav::detectKeypoints -> kd.corners float64[n, 2]."""
        img = self._image(img)
        h, w = img.shape
        cap = num_features if num_features > 0 else w * h
        out = np.zeros((max(cap, 1), 2), dtype=np.float64)
        n = _sz(0)
        _check(lib().vsm_detect_keypoints(self._h, C.c_void_p(img.ctypes.data), w, h, img.strides[0],
                                          int(num_features), _ptr(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def detect_and_describe(self, img, num_features: int, rotate_features: bool = True):
        """visnav::detectKeypointsAndDescriptors -> (corners f64[n,2], angles f64[n], descriptors u64[n,4])."""
        img = self._image(img)
        h, w = img.shape
        cap = max(num_features if num_features > 0 else w * h, 1)
        xy = np.zeros((cap, 2), dtype=np.float64)
        ang = np.zeros(cap, dtype=np.float64)
        desc = np.zeros((cap, 4), dtype=np.uint64)
        n = _sz(0)
        _check(lib().vsm_detect_and_describe(self._h, C.c_void_p(img.ctypes.data), w, h, img.strides[0],
                                             int(num_features), 1 if rotate_features else 0, _ptr(xy), _ptr(ang),
                                             _ptr(desc), cap, C.byref(n)))
        k = n.value
        return xy[:k].copy(), ang[:k].copy(), desc[:k].copy()

    def detect_and_describe_batch(self, images, num_features: int, rotate_features: bool = True,
                                  keep_as_database: bool = False, want=("corners", "angles", "descriptors")):
        """The loop of detect_keypoints(), sfm.cpp:1191-1201, in one call: `images` = equally sized
        uint8 images (a list of 2-D arrays or one 3-D array) -> (offsets uint32[n+1], corners f64[t,2],
        angles f64[t], descriptors u64[t,4]); arrays not named in `want` come back as None.
        keep_as_database: the descriptors also become the context's database (as after db_upload)."""
        imgs = [self._image(im) for im in images]
        n_img = len(imgs)
        if n_img == 0:
            h = w = pitch = 1
        else:
            h, w = imgs[0].shape
            pitch = imgs[0].strides[0]
            for im in imgs:
                if im.shape != (h, w) or im.strides[0] != pitch:
                    raise ValueError("the images of a batch must have one size and pitch")
        ptrs = (C.c_void_p * max(n_img, 1))(*[im.ctypes.data for im in imgs])
        offsets = np.zeros(n_img + 1, dtype=np.uint32)
        cap = n_img * (num_features if num_features > 0 else 4096)
        while True:
            xy = np.zeros((max(cap, 1), 2), dtype=np.float64) if "corners" in want else None
            ang = np.zeros(max(cap, 1), dtype=np.float64) if "angles" in want else None
            desc = np.zeros((max(cap, 1), 4), dtype=np.uint64) if "descriptors" in want else None
            rc = lib().vsm_detect_and_describe_batch(self._h, ptrs, n_img, w, h, pitch, int(num_features),
                                                     1 if rotate_features else 0, _ptr(offsets),
                                                     _ptr(xy) if xy is not None else None, _ptr(ang) if ang is not None else None,
                                                     _ptr(desc) if desc is not None else None, cap, 1 if keep_as_database else 0)
            if rc == VSM_ERR_CAPACITY and int(offsets[-1]) > cap:
                cap = int(offsets[-1])
                continue
            _check(rc)
            break
        t = int(offsets[-1])
        if keep_as_database:
            self._offsets = offsets.copy()
        return (offsets, xy[:t].copy() if xy is not None else None, ang[:t].copy() if ang is not None else None,
                desc[:t].copy() if desc is not None else None)

    def orb_descriptors(self, img, corners_xy, angles) -> np.ndarray:
        """visnav::computeDescriptors with caller-given angles (keypoints.h:203-237)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        xy = np.ascontiguousarray(corners_xy, dtype=np.float64).reshape(-1, 2)
        ang = np.ascontiguousarray(angles, dtype=np.float64)
        if len(ang) != len(xy):
            raise ValueError("angles and corners differ in length")
        desc = np.zeros((len(xy), 4), dtype=np.uint64)
        h, w = img.shape
        _check(lib().vsm_orb_descriptors(self._h, C.c_void_p(img.ctypes.data), w, h, img.strides[0], _ptr(xy),
                                         _ptr(ang), len(xy), _ptr(desc)))
        return desc

    # -- bow_voc.h:57-131 ----------------------------------------------------------
    def voc_upload(self, voc):
        """`voc`: dict with node_desc uint64[n,4], child_off uint32[n+1], child_idx uint32[*],
        node_id / word_id uint32[n], weight f64[n] (synth.vocabulary, cereal_io.load_vocabulary_binary)."""
        nd = as_descriptors(voc["node_desc"])
        a = {k: np.ascontiguousarray(voc[k], dtype=np.uint32) for k in ("child_off", "child_idx", "node_id", "word_id")}
        w = np.ascontiguousarray(voc["weight"], dtype=np.float64)
        n = len(nd)
        if len(a["child_off"]) != n + 1 or len(a["node_id"]) != n or len(a["word_id"]) != n or len(w) != n:
            raise ValueError("vocabulary arrays disagree in length")
        _check(lib().vsm_voc_upload(self._h, _ptr(nd), _ptr(a["child_off"]), _ptr(a["child_idx"]), _ptr(a["node_id"]),
                                    _ptr(a["word_id"]), _ptr(w), n))

    def bow_words(self, desc):
        """transformFeatureToWord for every descriptor -> (word uint32[n], weight f64[n])."""
        d = as_descriptors(desc)
        word = np.zeros(len(d), dtype=np.uint32)
        weight = np.zeros(len(d), dtype=np.float64)
        _check(lib().vsm_bow_words(self._h, _ptr(d), len(d), _ptr(word), _ptr(weight)))
        return word, weight

    def bow_transform(self, desc):
        """BowVocabulary::transform (bow_voc.h:94-131): [(word_id, value)] in order of first
        appearance, zero-weight words skipped, L1-normalised; words found on the GPU, the few
        hundred double additions done here in the reference's order."""
        return bow_vector(*self.bow_words(desc))

    # -- bow_db.h:45-168, sfm.cpp:1366-1392 -----------------------------------------
    @staticmethod
    def _bow_csr(vectors):
        """[[(word, value), ...], ...] or an (off, words, values) triple -> contiguous arrays."""
        if isinstance(vectors, tuple) and len(vectors) == 3:
            off, words, values = vectors
        else:
            off = np.zeros(len(vectors) + 1, dtype=np.uint32)
            off[1:] = np.cumsum([len(v) for v in vectors])
            words = [w for v in vectors for w, _ in v]
            values = [x for v in vectors for _, x in v]
        off = np.ascontiguousarray(off, dtype=np.uint32)
        words = np.ascontiguousarray(words, dtype=np.uint32)
        values = np.ascontiguousarray(values, dtype=np.float64)
        if off.ndim != 1 or len(off) < 1 or len(words) != len(values) or (len(off) and int(off[-1]) != len(words)):
            raise ValueError("BoW vector arrays disagree in length")
        return off, words, values

    def bowdb_upload(self, vectors):
        """BowDatabase::clear + insert (bow_db.h:49-61) of images 0 .. n-1 in this order; `vectors`:
        one BowVector [(word, value), ...] per image, or (off uint32[n+1], words, values)."""
        off, words, values = self._bow_csr(vectors)
        _check(lib().vsm_bowdb_upload(self._h, _ptr(off), _ptr(words), _ptr(values), len(off) - 1))
        self._bowdb_images = len(off) - 1

    def bowdb_query(self, vectors, num_results, db_limit=None):
        """BowDatabase::query (bow_db.h:63-131) for every vector -> list of (index int32[k], score
        float64[k]) ascending by score (equal scores: lower image index first).  db_limit: per query,
        how many of the first images it sees."""
        off, words, values = self._bow_csr(vectors)
        n_q, nr = len(off) - 1, int(num_results)
        idx = np.full((n_q, nr), -1, dtype=np.int32)
        score = np.zeros((n_q, nr), dtype=np.float64)
        cnt = np.zeros(n_q, dtype=np.uint32)
        lim = None if db_limit is None else np.ascontiguousarray(db_limit, dtype=np.uint32)
        if lim is not None and len(lim) != n_q:
            raise ValueError("one db_limit per query")
        _check(lib().vsm_bowdb_query(self._h, _ptr(off), _ptr(words), _ptr(values), n_q,
                                     None if lim is None else _ptr(lim), nr, _ptr(idx), _ptr(score), _ptr(cnt)))
        return [(idx[q, :cnt[q]].copy(), score[q, :cnt[q]].copy()) for q in range(n_q)]

    def bow_select_pairs(self, frame_ids, num_candidates=25):
        """The pair selection of match_bow (sfm.cpp:1366-1392; num_bow_candidates = 25, sfm.cpp:209) over
        the uploaded images -> uint32[n, 2] of (image, earlier image), ready for match_pairs."""
        fr = np.ascontiguousarray(frame_ids, dtype=np.int64)
        if len(fr) != getattr(self, "_bowdb_images", -1):
            raise ValueError("one frame id per uploaded image")
        cap = max(1, len(fr) * min(int(num_candidates), max(len(fr), 1)))
        pairs = np.zeros((cap, 2), dtype=np.uint32)
        n = _sz(0)
        _check(lib().vsm_bow_select_pairs(self._h, _ptr(fr), int(num_candidates), _ptr(pairs), cap, C.byref(n)))
        return pairs[:n.value].copy()

    # -- measurement ------------------------------------------------------------
    def set_profiling(self, on: bool):
        _check(lib().vsm_set_profiling(self._h, 1 if on else 0))

    def set_variant(self, v: int):
        _check(lib().vsm_set_variant(self._h, int(v)))

    def last_scan_stats(self, kind: int | None = None):
        """kind None = all scan launches of the last call, 0 = full/forward, 1 = pruned backward."""
        ms, n, comps = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        if kind is None:
            _check(lib().vsm_last_scan_stats(self._h, C.byref(ms), C.byref(n), C.byref(comps)))
        else:
            _check(lib().vsm_last_scan_stats_kind(self._h, int(kind), C.byref(ms), C.byref(n), C.byref(comps)))
        return {"scan_ms": ms.value, "launches": n.value, "comparisons": comps.value}

    def last_scan_kernel(self, kind: int = 0) -> int:
        """0 none yet, 1 popcount scan, 2 matrix cores 32x32x64, 3 matrix cores 16x16x128 (vsm_last_scan_kernel)."""
        return int(lib().vsm_last_scan_kernel(self._h, int(kind)))


class MultiMatcher:
    """match_all over several contexts from one process (vsm_db_upload_multi / vsm_match_pairs_multi):
    one context per entry of `devices` (a device may be listed more than once)."""

    def __init__(self, devices):
        self.matchers = [Matcher(d) for d in devices]
        self._arr = (_vp * len(self.matchers))(*[m._h for m in self.matchers])
        self._offsets = None

    def close(self):
        for m in self.matchers:
            m.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def db_upload(self, desc, offsets):
        desc = as_descriptors(desc)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        if offsets.ndim != 1 or len(offsets) < 1 or int(offsets[-1]) != len(desc):
            raise ValueError("offsets must be uint32[n_img+1] ending at len(desc)")
        _check(lib().vsm_db_upload_multi(self._arr, len(self.matchers), _ptr(desc), _ptr(offsets), len(offsets) - 1))
        self._offsets = offsets

    def match_pairs(self, pairs, threshold: int = 70, dist_2_best: float = 1.2, cap: int | None = None):
        """-> (offsets uint64[n+1], matches int32[total, 2]) in the order of `pairs`."""
        if self._offsets is None:
            raise VsmError(VSM_ERR_INVALID, "no database uploaded")
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        n = len(pairs)
        if cap is None:
            sizes = np.diff(self._offsets.astype(np.int64))
            ok = (pairs < len(sizes)).all(axis=1) if n else np.zeros(0, bool)
            cap = int(np.minimum(sizes[pairs[ok, 0]], sizes[pairs[ok, 1]]).sum()) if n else 0
        cap = max(int(cap), 1)
        offs = np.empty(n + 1, dtype=np.uint64)
        out = np.empty((cap, 2), dtype=np.int32)
        total = _sz(0)
        _check(lib().vsm_match_pairs_multi(self._arr, len(self.matchers), _ptr(pairs), n, int(threshold), float(dist_2_best),
                                           _ptr(offs), _ptr(out), cap, C.byref(total)))
        return offs, out[: total.value].copy()


def bow_vector(words, weights):
    """bow_voc.h:107-130 given each feature's (word, weight)."""
    order, acc, total = [], {}, 0.0
    for w, v in zip(np.asarray(words).tolist(), np.asarray(weights, dtype=np.float64).tolist()):
        if v == 0:
            continue
        total += v
        if w in acc:
            acc[w] += v
        else:
            acc[w] = v
            order.append(w)
    return [(w, acc[w] / total) for w in order]


def bow_vectors_of_images(words, weights, offsets):
    """bow_voc.h:107-130 for every image of a database at once: features offsets[i] .. offsets[i+1]
    belong to image i.  -> (vec_off uint32[n + 1], vec_words uint32[*], vec_values float64[*]), the
    `vsm_bowdb_upload` layout.  Same additions in the same order as `bow_vector` (np.add.at and
    np.cumsum add one element after the other), so the same bits."""
    words = np.asarray(words, dtype=np.uint32)
    weights = np.asarray(weights, dtype=np.float64)
    offsets = np.asarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    vec_off = np.zeros(n + 1, dtype=np.uint32)
    out_w, out_v = [], []
    for i in range(n):
        w, x = words[offsets[i]:offsets[i + 1]], weights[offsets[i]:offsets[i + 1]]
        keep = x != 0  # bow_voc.h:114
        w, x = w[keep], x[keep]
        if len(w):
            uniq, first, inv = np.unique(w, return_index=True, return_inverse=True)
            order = np.argsort(first, kind="stable")  # words in order of first appearance
            slot = np.empty(len(uniq), dtype=np.int64)
            slot[order] = np.arange(len(uniq))
            acc = np.zeros(len(uniq), dtype=np.float64)
            np.add.at(acc, slot[inv], x)
            total = np.cumsum(x)[-1]
            out_w.append(uniq[order].astype(np.uint32))
            out_v.append(acc / total)
        vec_off[i + 1] = vec_off[i] + (len(out_w[-1]) if len(w) else 0)
    return vec_off, (np.concatenate(out_w) if out_w else np.zeros(0, np.uint32)), \
        (np.concatenate(out_v) if out_v else np.zeros(0, np.float64))


_default = threading.local()


def match_descriptors(corner_descriptors_1, corner_descriptors_2, threshold: int, dist_2_best: float,
                      device: int = 0) -> np.ndarray:
    """Drop-in for visnav::matchDescriptors (keypoints.h:247-250): returns the
    `matches` vector as int32[n, 2].  Uses one context per calling thread, so it
    is re-entrant the way the reference's callers need (sfm.cpp:1295)."""
    m = getattr(_default, "m", None)
    if m is None or m.device != device:
        m = Matcher(device)
        _default.m = m
    return m.match(corner_descriptors_1, corner_descriptors_2, threshold, dist_2_best)
