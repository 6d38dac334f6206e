"""Host-side mirror of the reference's src/features interfaces over the C ABI (include/orbgpu.h).

Names, argument meaning and error behaviour follow the reference so that the parity tests read like its
own:  IFeatureExtractor / ORBFeatureExtractor (src/features/ifeature_extractor.h:24-60,
orb_feature_extractor.h:22-45), Features (features.h:23-54), SNNMatcher
(matching/second_nearest_neighbor_matcher.hpp:19-84) with its AreaTo / ProjectionSearch candidate
generators, OrientationValidator (matching/validators/orientation_validator.h:19-32).
All compute goes through liborbgpu.so; nothing here falls back to the CPU.
"""
import ctypes as C
import math
import struct

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, OrbGpuError

ORBFE = 0  # FeatureExtractorType::ORBFE, feature_extractor_type.h:9-12


class Features:
    """features::Features (features.h:23-54): descriptors N x 32 u8 + keypoints (+ the per-keypoint geometry the
    feature-handler factory fills afterwards, factories/feature_handler_factory.cpp:53-60)."""

    def __init__(self, width, height):
        self.image_width = float(width)
        self.image_height = float(height)
        self.descriptors = np.zeros((0, 32), np.uint8)
        self.keypoints = np.zeros(0, KP_DTYPE)
        self.undistorted_keypoints = np.zeros((0, 2), np.float64)
        self.undistorted_and_unprojected_keypoints = np.zeros((0, 3), np.float64)

    def Size(self):
        return len(self.keypoints)

    # -- wire format of operator<< / Features(std::istream &), features.cpp:14-79 (little endian, x86-64):
    #    image_width, image_height (precision_t = f64); descriptor cols, rows (Eigen::Index = i64); rows x cols bytes;
    #    per keypoint: level i32, size f64, angle f64, pt 2 x f64; then rows x 2 f64 undistorted and rows x 3 f64
    #    unprojected points.  The reader consumes `rows` entries of each array, so all of them are written.
    def Serialize(self):
        n = self.Size()
        out = [np.array([self.image_width, self.image_height], "<f8").tobytes(),
               np.array([self.descriptors.shape[1] if n else 32, n], "<i8").tobytes(),
               np.ascontiguousarray(self.descriptors, np.uint8).tobytes()]
        rec = np.zeros(n, np.dtype([("level", "<i4"), ("size", "<f8"), ("angle", "<f8"), ("x", "<f8"), ("y", "<f8")]))
        rec["level"], rec["size"] = self.keypoints["level"], self.keypoints["size"]
        rec["angle"], rec["x"], rec["y"] = self.keypoints["angle"], self.keypoints["x"], self.keypoints["y"]
        out.append(rec.tobytes())
        for arr, dim in ((self.undistorted_keypoints, 2), (self.undistorted_and_unprojected_keypoints, 3)):
            a = np.zeros((n, dim), "<f8")
            a[:len(arr)] = np.asarray(arr, np.float64).reshape(-1, dim)[:n]
            out.append(a.tobytes())
        return b"".join(out)

    @classmethod
    def Deserialize(cls, data):
        w, h = np.frombuffer(data, "<f8", 2, 0)
        cols, rows = (int(v) for v in np.frombuffer(data, "<i8", 2, 16))
        f = cls(w, h)
        pos = 32
        f.descriptors = np.frombuffer(data, np.uint8, rows * cols, pos).reshape(rows, cols).copy()
        pos += rows * cols
        rec = np.frombuffer(data, np.dtype([("level", "<i4"), ("size", "<f8"), ("angle", "<f8"), ("x", "<f8"), ("y", "<f8")]),
                            rows, pos)
        pos += rows * 36
        f.keypoints = np.zeros(rows, KP_DTYPE)
        for name in ("level", "size", "angle", "x", "y"):
            f.keypoints[name] = rec[name]
        f.undistorted_keypoints = np.frombuffer(data, "<f8", rows * 2, pos).reshape(rows, 2).copy()
        pos += rows * 16
        f.undistorted_and_unprojected_keypoints = np.frombuffer(data, "<f8", rows * 3, pos).reshape(rows, 3).copy()
        pos += rows * 24
        assert pos == len(data), "trailing bytes"
        return f


class ORBFeatureExtractor:
    """ORBFeatureExtractor(image_width, image_height, scale_factor, levels, init_threshold_FAST,
    min_threshold_FAST) -- orb_feature_extractor.h:24-29 -- backed by an orbgpu context."""

    def __init__(self, image_width, image_height, scale_factor, levels, init_threshold_FAST, min_threshold_FAST,
                 device=0, max_batch=1, max_features=2000):
        self._lib = _lib.load()
        self._p = _lib.Params(int(image_width), int(image_height), float(scale_factor), int(levels),
                              int(init_threshold_FAST), int(min_threshold_FAST), int(device), int(max_batch),
                              int(max_features))
        h = C.c_void_p()
        rc = self._lib.orbgpu_create(C.byref(self._p), C.byref(h))
        if rc != _lib.OK:
            raise OrbGpuError(rc, self._lib.orbgpu_last_error(None).decode())
        self._h = h
        self.image_width_, self.image_height_ = int(image_width), int(image_height)
        self.scale_factor_ = float(scale_factor)
        self.log_scale_factor_ = math.log(self.scale_factor_)
        self.max_batch, self.max_features = int(max_batch), int(max_features)
        g = self.geometry(max_features)
        self.scale_factors_ = [g.scale[i] for i in range(levels)]
        self.level_sigma2_ = [g.sigma2[i] for i in range(levels)]

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.orbgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != _lib.OK:
            raise OrbGpuError(rc, self._lib.orbgpu_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    def geometry(self, feature_count):
        g = _lib.Geometry()
        self._check(self._lib.orbgpu_get_geometry(self._h, int(feature_count), C.byref(g)))
        return g

    # -- IFeatureExtractor -----------------------------------------------------------------------
    def Extract(self, image, out_features, feature_count):
        """Extract(image, out_features, feature_count) -> int.  Returns -1 on an empty image like
        orb_feature_extractor.cpp:533, otherwise 0 (the reference's monoIndex, App. B.4)."""
        if image is None or image.size == 0:
            return -1
        image = np.ascontiguousarray(image, np.uint8)
        cap = self.geometry(feature_count).keypoint_capacity
        kp = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        rc = self._lib.orbgpu_extract(self._h, _lib.ptr(image), image.shape[1], image.shape[0], image.strides[0],
                                      int(feature_count), _lib.ptr(kp), _lib.ptr(desc), cap, C.byref(n))
        if rc == _lib.E_EMPTY:
            return -1
        self._check(rc)
        out_features.keypoints = kp[:n.value].copy()
        out_features.descriptors = desc[:n.value].copy()
        return 0

    def ExtractBatch(self, images, feature_count):
        """n frames in one pass (orbgpu_extract_batch). Returns a list of Features."""
        images = [np.ascontiguousarray(im, np.uint8) for im in images]
        n_frames = len(images)
        strides = {im.strides[0] for im in images}
        assert len(strides) == 1, "all frames must share one row stride"
        cap = self.geometry(feature_count).keypoint_capacity
        kp = np.zeros((n_frames, cap), KP_DTYPE)
        desc = np.zeros((n_frames, cap, 32), np.uint8)
        n = np.zeros(n_frames, np.int32)
        ptrs = (C.c_void_p * n_frames)(*[im.ctypes.data for im in images])
        self._check(self._lib.orbgpu_extract_batch(self._h, ptrs, n_frames, strides.pop(), int(feature_count),
                                                   _lib.ptr(kp), _lib.ptr(desc), cap, _lib.ptr(n)))
        out = []
        for f in range(n_frames):
            ft = Features(self.image_width_, self.image_height_)
            ft.keypoints = kp[f, :n[f]].copy()
            ft.descriptors = desc[f, :n[f]].copy()
            out.append(ft)
        return out

    def debug_dump(self, frame=0):
        """Stage intermediates of the last Extract call (orbgpu_debug_dump)."""
        g = self.geometry(self.max_features)
        tot = int(g.pyramid_pixels)
        pyr, score, blur = (np.zeros(tot, np.uint8) for _ in range(3))
        cap = tot // 4 + 1024
        cx, cy, cs = (np.zeros(cap, np.int32) for _ in range(3))
        ncl = np.zeros(_lib.MAX_LEVELS, np.int32)
        self._check(self._lib.orbgpu_debug_dump(self._h, int(frame), _lib.ptr(pyr), _lib.ptr(score), _lib.ptr(blur),
                                                _lib.ptr(cx), _lib.ptr(cy), _lib.ptr(cs), cap, _lib.ptr(ncl)))
        n = int(ncl[:g.levels].sum())
        return {"pyramid": pyr, "score": score, "blurred": blur, "cand_x": cx[:n], "cand_y": cy[:n],
                "cand_score": cs[:n], "n_cands": ncl[:g.levels].tolist(), "geometry": g}

    def set_stage_timing(self, all_stages):
        self._check(self._lib.orbgpu_set_stage_timing(self._h, 1 if all_stages else 0))

    def stage_times(self):
        ms = np.zeros(8, np.float32)
        self._check(self._lib.orbgpu_stage_times(self._h, _lib.ptr(ms)))
        return dict(zip(("pyramid", "fast_score", "orient", "quadtree", "blur", "describe", "total", "fast_level0"),
                        ms.tolist()))

    def popcount_peak(self):
        v = C.c_double(0)
        self._check(self._lib.orbgpu_popcount_peak(self._h, C.byref(v)))
        return v.value

    def Type(self):
        return ORBFE

    def GetAcceptableSquareError(self, level):
        # orb_feature_extractor.cpp:654-660
        if level >= len(self.level_sigma2_):
            return 0
        return self.level_sigma2_[level]

    def ComputeInvariantDistances(self, point, key_point_level):
        # :854-862 -> (max_distance, min_distance)
        distance = math.sqrt(sum(float(v) * float(v) for v in point))
        out_max = distance * self.scale_factors_[key_point_level]
        return out_max, out_max / self.scale_factors_[-1]

    def PredictScale(self, distance, max_distance):
        # :864-873
        ratio = max_distance / distance
        scale = int(max(0.0, math.ceil(math.log(ratio) / self.log_scale_factor_)))
        return min(scale, len(self.scale_factors_) - 1)

    def ComputeDistance(self, d1, d2):
        # :83-98; host-side like the reference's callers (orbgpu_hamming256 is the C inline twin)
        x = np.bitwise_xor(np.asarray(d1, np.uint8).reshape(-1), np.asarray(d2, np.uint8).reshape(-1))
        return int(np.unpackbits(x).sum())

    def GetScaleFactors(self):
        return self.scale_factors_

    def GetHighThreshold(self):
        return 100  # :75-77

    def GetLowThreshold(self):
        return 50  # :79-81

    def Serialize(self):
        # :875-885: width, height (unsigned), scale_factor (double), levels (size_t), two unsigned thresholds
        return struct.pack("<IIdQII", self.image_width_, self.image_height_, self.scale_factor_,
                           len(self.scale_factors_), self._p.fast_init, self._p.fast_min)


class SNNMatcher:
    """SNNMatcher(nearest_neighbour_ratio, threshold) -- second_nearest_neighbor_matcher.hpp:26-28.
    The three Match* methods are MatchWithIterators driven by the reference's three candidate generators;
    each returns {to_id: from_id} like MatchMapType."""

    def __init__(self, nearest_neighbour_ratio, threshold, feature_extractor):
        self.nearest_neighbour_ratio_ = float(nearest_neighbour_ratio)
        self.THRESHOLD_ = int(threshold)
        self._fe = feature_extractor
        self._lib = feature_extractor._lib

    def _run(self, args, n_to, keep):
        idx = np.full(max(n_to, 1), -1, np.int32)
        dist = np.zeros(max(n_to, 1), np.uint32)
        self._fe._check(self._lib.orbgpu_match_window(self._fe.handle, C.byref(args), _lib.ptr(idx), _lib.ptr(dist)))
        del keep
        return idx[:n_to], dist[:n_to]

    def MatchRaw(self, to_desc, from_desc, to_xy=None, to_window=None, to_min_level=None, to_max_level=None,
                 to_active=None, from_xy=None, from_level=None):
        to_desc = np.ascontiguousarray(to_desc, np.uint8).reshape(-1, 32)
        from_desc = np.ascontiguousarray(from_desc, np.uint8).reshape(-1, 32)
        keep = [to_desc, from_desc]

        def prep(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return a

        to_xy, from_xy = prep(to_xy, np.float64), prep(from_xy, np.float64)
        to_window = prep(to_window, np.uint64)
        to_min_level, to_max_level = prep(to_min_level, np.int32), prep(to_max_level, np.int32)
        to_active, from_level = prep(to_active, np.uint8), prep(from_level, np.int32)
        a = _lib.WindowArgs(_lib.ptr(to_desc), _lib.ptr(to_xy), _lib.ptr(to_window), _lib.ptr(to_min_level),
                            _lib.ptr(to_max_level), _lib.ptr(to_active), len(to_desc), _lib.ptr(from_desc),
                            _lib.ptr(from_xy), _lib.ptr(from_level), len(from_desc), self.THRESHOLD_,
                            self.nearest_neighbour_ratio_)
        return self._run(a, len(to_desc), keep)

    def MatchBruteForce(self, to_desc, from_desc):
        idx, _ = self.MatchRaw(to_desc, from_desc)
        return {i: int(j) for i, j in enumerate(idx) if j >= 0}

    def MatchArea(self, features_to, features_from, window_size):
        """AreaToIterator(0 .. Size(), features_to, features_from, window_size): window search, level 0 on both
        sides, id 0 visited whatever its level (area_to_iterator.h:24-43, area_to_pointee.h:43-54)."""
        kt, kf = features_to.keypoints, features_from.keypoints
        n = len(kt)
        active = (kt["level"] == 0).astype(np.uint8)
        if n:
            active[0] = 1
        idx, _ = self.MatchRaw(features_to.descriptors, features_from.descriptors,
                               to_xy=np.stack([kt["x"], kt["y"]], 1), to_window=np.full(n, int(window_size), np.uint64),
                               to_min_level=np.zeros(n, np.int32), to_max_level=np.zeros(n, np.int32), to_active=active,
                               from_xy=np.stack([kf["x"], kf["y"]], 1), from_level=kf["level"])
        return {i: int(j) for i, j in enumerate(idx) if j >= 0}

    def MatchProjection(self, visibles, features_from):
        """ProjectionSearchIterator over MapPointVisibilityParams-like records
        (frame/map_point_visibility_params.h:16-22): dicts with descriptor, position (x, y), window_size
        (double, truncated like features.cpp:82) and level; 'to' ids are positions in the list."""
        n = len(visibles)
        kf = features_from.keypoints
        if n == 0:
            return {}
        desc = np.stack([np.asarray(v["descriptor"], np.uint8) for v in visibles])
        xy = np.array([v["position"] for v in visibles], np.float64).reshape(n, 2)
        win = np.array([int(v["window_size"]) for v in visibles], np.uint64)  # size_t truncation
        lv = np.array([v["level"] for v in visibles], np.int32)
        idx, _ = self.MatchRaw(desc, features_from.descriptors, to_xy=xy, to_window=win,
                               to_min_level=np.maximum(lv - 1, 0), to_max_level=lv + 1,
                               from_xy=np.stack([kf["x"], kf["y"]], 1), from_level=kf["level"])
        return {i: int(j) for i, j in enumerate(idx) if j >= 0}

    @staticmethod
    def JoinFeatureVectors(fv_to, fv_from):
        """Merge join of two DBoW2::FeatureVector maps ({node id: [feature ids]}) the way
        BowToIterator::AdvanceUntilSameNode walks them (bow_to_iterator.cpp:27-47): nodes present in both, in node
        order.  Returns (to_off, to_ids, from_off, from_ids) for orbgpu_match_groups."""
        to_off, from_off, to_ids, from_ids = [0], [0], [], []
        for node in sorted(set(fv_to) & set(fv_from)):
            to_ids.extend(fv_to[node])
            from_ids.extend(fv_from[node])
            to_off.append(len(to_ids))
            from_off.append(len(from_ids))
        return (np.array(to_off, np.int32), np.array(to_ids, np.uint32), np.array(from_off, np.int32),
                np.array(from_ids, np.uint32))

    def MatchBow(self, to_desc, from_desc, fv_to, fv_from):
        """SNNMatcher<BowToIterator>: the matcher of DBoW2Handler::FastMatch (dbo_w2_handler.cpp:25-72); candidates
        of a 'to' feature are the 'from' features of the same vocabulary node."""
        to_desc = np.ascontiguousarray(to_desc, np.uint8).reshape(-1, 32)
        from_desc = np.ascontiguousarray(from_desc, np.uint8).reshape(-1, 32)
        to_off, to_ids, from_off, from_ids = self.JoinFeatureVectors(fv_to, fv_from)
        n_to = len(to_desc)
        a = _lib.GroupArgs(_lib.ptr(to_desc), n_to, _lib.ptr(from_desc), len(from_desc), len(to_off) - 1,
                           _lib.ptr(to_off), _lib.ptr(to_ids), _lib.ptr(from_off), _lib.ptr(from_ids), self.THRESHOLD_,
                           self.nearest_neighbour_ratio_)
        idx = np.full(max(n_to, 1), -1, np.int32)
        dist = np.zeros(max(n_to, 1), np.uint32)
        self._fe._check(self._lib.orbgpu_match_groups(self._fe.handle, C.byref(a), _lib.ptr(idx), _lib.ptr(dist)))
        return {i: int(j) for i, j in enumerate(idx[:n_to]) if j >= 0}


class OrientationValidator:
    """OrientationValidator(kp_to, kp_from).Validate(inout_matches_12) -> number discarded
    (validators/orientation_validator.cpp:11-40)."""

    def __init__(self, kp_to, kp_from, feature_extractor):
        self.kp_to_, self.kp_from_ = kp_to, kp_from
        self._fe = feature_extractor

    def Validate(self, inout_matches_12):
        n_to, n_from = len(self.kp_to_), len(self.kp_from_)
        m = np.full(max(n_to, 1), -1, np.int32)
        for i, j in inout_matches_12.items():
            m[i] = j
        a_to = np.ascontiguousarray(self.kp_to_["angle"], np.float32)
        a_from = np.ascontiguousarray(self.kp_from_["angle"], np.float32)
        nd = C.c_int32(0)
        self._fe._check(self._fe._lib.orbgpu_rotation_filter(self._fe.handle, _lib.ptr(a_to), n_to, _lib.ptr(a_from),
                                                             n_from, _lib.ptr(m), C.byref(nd)))
        for i in list(inout_matches_12):
            if m[i] < 0:
                del inout_matches_12[i]
        return nd.value


def ComputeDistinctiveDescriptors(feature_extractor, descriptor_groups):
    """MapPoint::ComputeDistinctiveDescriptor (map/map_point.cpp:135-171) for many map points in one call.
    descriptor_groups: one (N_p x 32) uint8 array per map point (its observations' descriptors in
    AppendDescriptorsToList order).  Returns per point the index of the chosen descriptor, or -1 if it has none."""
    groups = [np.ascontiguousarray(g, np.uint8).reshape(-1, 32) for g in descriptor_groups]
    offsets = np.zeros(len(groups) + 1, np.int32)
    for p, g in enumerate(groups):
        offsets[p + 1] = offsets[p] + len(g)
    desc = np.concatenate(groups) if groups and offsets[-1] else np.zeros((0, 32), np.uint8)
    best = np.full(max(len(groups), 1), -1, np.int32)
    feature_extractor._check(feature_extractor._lib.orbgpu_distinctive_descriptor(
        feature_extractor.handle, _lib.ptr(desc), _lib.ptr(offsets), len(groups), _lib.ptr(best)))
    return best[:len(groups)]


class MonocularCamera:
    """camera::MonocularCamera (camera/monocular_camera.h) as far as the feature path uses it: intrinsics plus an
    optional distortion model, and the per-keypoint loop of FeatureHandlerFactory::Create
    (factories/feature_handler_factory.cpp:53-60) as one call."""

    def __init__(self, fx, fy, cx, cy, model=_lib.DISTORTION_NONE, k=()):
        self._c = _lib.Camera(float(fx), float(fy), float(cx), float(cy), int(model), 0)
        for i, v in enumerate(k):
            self._c.k[i] = float(v)

    def UndistortFeatures(self, feature_extractor, features):
        """fills features.undistorted_keypoints / undistorted_and_unprojected_keypoints; returns the ok flags
        (bit 0: UndistortPoint, bit 1: UnprojectAndUndistort)"""
        n = features.Size()
        kp = np.ascontiguousarray(features.keypoints)
        und = np.zeros((max(n, 1), 2), np.float64)
        unp = np.zeros((max(n, 1), 3), np.float64)
        ok = np.zeros(max(n, 1), np.uint8)
        feature_extractor._check(feature_extractor._lib.orbgpu_keypoint_geometry(
            feature_extractor.handle, C.byref(self._c), _lib.ptr(kp), n, _lib.ptr(und), _lib.ptr(unp), _lib.ptr(ok)))
        features.undistorted_keypoints = und[:n]
        features.undistorted_and_unprojected_keypoints = unp[:n]
        return ok[:n]
