"""ctypes loader for liborbgpu.so (the C ABI of include/orbgpu.h).

There is no CPU fallback: if the HIP library is missing or no GPU is usable this module raises.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "liborbgpu.so"

MAX_LEVELS = 16
OK, E_EMPTY, E_INVALID, E_HIP, E_CAPACITY = 0, -1, -2, -3, -4


class OrbGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"orbgpu error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("scale_factor", C.c_double), ("levels", C.c_int32),
                ("fast_init", C.c_int32), ("fast_min", C.c_int32), ("device", C.c_int32), ("max_batch", C.c_int32),
                ("max_features", C.c_int32)]


class Geometry(C.Structure):
    _fields_ = [("levels", C.c_int32)] + \
               [(n, C.c_int32 * MAX_LEVELS) for n in ("w", "h", "budget", "ncols", "nrows", "wcell", "hcell", "kp_size")] + \
               [(n, C.c_double * MAX_LEVELS) for n in ("scale", "inv_scale", "sigma2")] + \
               [("pyramid_pixels", C.c_int64), ("keypoint_capacity", C.c_int32)]


class WindowArgs(C.Structure):
    _fields_ = [("to_desc", C.c_void_p), ("to_xy", C.c_void_p), ("to_window", C.c_void_p), ("to_min_level", C.c_void_p),
                ("to_max_level", C.c_void_p), ("to_active", C.c_void_p), ("n_to", C.c_int32), ("from_desc", C.c_void_p),
                ("from_xy", C.c_void_p), ("from_level", C.c_void_p), ("n_from", C.c_int32), ("threshold", C.c_uint32),
                ("ratio", C.c_double)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("model", C.c_int32),
                ("reserved", C.c_int32), ("k", C.c_double * 8)]


DISTORTION_NONE, DISTORTION_BARREL5, DISTORTION_BARREL8, DISTORTION_FISHEYE = -1, 0, 1, 2


class GroupArgs(C.Structure):
    _fields_ = [("to_desc", C.c_void_p), ("n_to", C.c_int32), ("from_desc", C.c_void_p), ("n_from", C.c_int32),
                ("n_groups", C.c_int32), ("to_off", C.c_void_p), ("to_ids", C.c_void_p), ("from_off", C.c_void_p),
                ("from_ids", C.c_void_p), ("threshold", C.c_uint32), ("ratio", C.c_double)]


class BowScoreArgs(C.Structure):
    _fields_ = [("q_words", C.c_void_p), ("q_values", C.c_void_p), ("q_counts", C.c_void_p), ("n_q", C.c_int32),
                ("n_keyframes", C.c_int32), ("kf_offsets", C.c_void_p), ("kf_words", C.c_void_p), ("kf_values", C.c_void_p),
                ("kf_counts", C.c_void_p), ("scoring", C.c_int32), ("reserved", C.c_int32)]


ENGINE_POPCOUNT, ENGINE_MFMA_I8 = 0, 1
SEGMENT_DTYPE = np.dtype([("to_begin", "<i4"), ("to_count", "<i4"), ("from_begin", "<i4"), ("from_count", "<i4")])
KP_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("angle", "<f4"), ("size", "<i4"), ("level", "<i4"), ("reserved", "<i4")])
assert KP_DTYPE.itemsize == 32

# every symbol include/orbgpu.h declares (the inline orbgpu_hamming256 aside)
EXPORTS = ["orbgpu_create", "orbgpu_destroy", "orbgpu_last_error", "orbgpu_get_geometry", "orbgpu_stream",
           "orbgpu_synchronize", "orbgpu_trim", "orbgpu_set_input_ready", "orbgpu_set_stream", "orbgpu_set_batch_slices", "orbgpu_wait_event", "orbgpu_record_event", "orbgpu_extract", "orbgpu_extract_batch", "orbgpu_extract_batch_device",
           "orbgpu_debug_dump", "orbgpu_match_bruteforce", "orbgpu_match_bruteforce_device", "orbgpu_match_bruteforce_noalloc_device", "orbgpu_set_bruteforce_engine",
           "orbgpu_match_window",
           "orbgpu_match_window_device", "orbgpu_match_window_segments_device", "orbgpu_grid_build_device",
           "orbgpu_match_window_grid_segments_device", "orbgpu_match_frames_device", "orbgpu_match_groups", "orbgpu_match_groups_device",
           "orbgpu_match_allpairs_device", "orbgpu_match_allpairs_angles_device", "orbgpu_vocab_create", "orbgpu_vocab_destroy", "orbgpu_bow_transform",
           "orbgpu_bow_transform_device", "orbgpu_distinctive_descriptor", "orbgpu_distinctive_descriptor_device",
           "orbgpu_keypoint_geometry", "orbgpu_keypoint_geometry_device", "orbgpu_bow_score", "orbgpu_bow_score_device",
           "orbgpu_bowdb_create", "orbgpu_bowdb_destroy", "orbgpu_bowdb_append", "orbgpu_bowdb_erase", "orbgpu_bowdb_size",
           "orbgpu_bowdb_query",
           "orbgpu_rotation_filter", "orbgpu_stage_times", "orbgpu_set_stage_timing",
           "orbgpu_popcount_peak"]

_lib = None


def load():
    """Load liborbgpu.so; raises if it has not been built (python __graft_entry__.py builds it)."""
    global _lib
    if _lib is not None:
        return _lib
    # ORBGPU_LIBRARY: another build of the same library (A/B timing of two builds on one box); default: the in-tree one
    path = Path(os.environ.get("ORBGPU_LIBRARY") or LIB_PATH)
    if not path.exists():
        raise OrbGpuError(E_HIP, f"{path} is missing: build it with `make -C orb_slam3_amd/csrc` "
                                 "(there is no CPU fallback)")
    lib = C.CDLL(str(path))
    vp, i32, u32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_uint32, C.c_int64, C.c_uint64, C.c_double
    sig = {
        "orbgpu_create": (i32, [C.POINTER(Params), C.POINTER(vp)]),
        "orbgpu_destroy": (None, [vp]),
        "orbgpu_last_error": (C.c_char_p, [vp]),
        "orbgpu_get_geometry": (i32, [vp, i32, C.POINTER(Geometry)]),
        "orbgpu_stream": (vp, [vp]),
        "orbgpu_synchronize": (i32, [vp]),
        "orbgpu_trim": (i32, [vp]),
        "orbgpu_set_input_ready": (i32, [vp, i32, vp]),
        "orbgpu_set_stream": (i32, [vp, vp]),
        "orbgpu_set_batch_slices": (i32, [vp, i32]),
        "orbgpu_wait_event": (i32, [vp, vp]),
        "orbgpu_record_event": (i32, [vp, vp]),
        "orbgpu_extract": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, i32, C.POINTER(i32)]),
        "orbgpu_extract_batch": (i32, [vp, C.POINTER(vp), i32, i32, i32, vp, vp, i32, vp]),
        "orbgpu_extract_batch_device": (i32, [vp, vp, i64, i32, i32, i32, vp, vp, i32, vp]),
        "orbgpu_debug_dump": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, i32, vp]),
        "orbgpu_match_bruteforce": (i32, [vp, vp, i32, vp, i32, u32, dbl, vp, vp]),
        "orbgpu_match_bruteforce_device": (i32, [vp, vp, i32, vp, i32, u32, dbl, vp, vp]),
        "orbgpu_match_bruteforce_noalloc_device": (i32, [vp, vp, i32, vp, i32, u32, dbl, vp, vp]),
        "orbgpu_set_bruteforce_engine": (i32, [vp, i32]),
        "orbgpu_match_window": (i32, [vp, C.POINTER(WindowArgs), vp, vp]),
        "orbgpu_match_window_device": (i32, [vp, C.POINTER(WindowArgs), vp, vp]),
        "orbgpu_match_window_segments_device": (i32, [vp, C.POINTER(WindowArgs), vp, i32, i32, vp, vp]),
        "orbgpu_grid_build_device": (i32, [vp, vp, vp, i32, i32, dbl, dbl, vp, vp]),
        "orbgpu_match_window_grid_segments_device": (i32, [vp, C.POINTER(WindowArgs), vp, i32, i32, i32, dbl, dbl, vp, vp]),
        "orbgpu_match_frames_device": (i32, [vp, vp, vp, vp, i32, vp, vp, i32, u64, u32, dbl, i32, vp, vp, vp]),
        "orbgpu_match_groups": (i32, [vp, C.POINTER(GroupArgs), vp, vp]),
        "orbgpu_match_groups_device": (i32, [vp, C.POINTER(GroupArgs), i32, vp, vp]),
        "orbgpu_match_allpairs_device": (i32, [vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, u32, dbl, i32, vp, vp]),
        "orbgpu_match_allpairs_angles_device": (i32, [vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, u32, dbl, i32, vp, vp]),
        "orbgpu_distinctive_descriptor": (i32, [vp, vp, vp, i32, vp]),
        "orbgpu_distinctive_descriptor_device": (i32, [vp, vp, vp, i32, vp]),
        "orbgpu_keypoint_geometry": (i32, [vp, C.POINTER(Camera), vp, i32, vp, vp, vp]),
        "orbgpu_keypoint_geometry_device": (i32, [vp, C.POINTER(Camera), vp, vp, i32, i32, vp, vp, vp]),
        "orbgpu_vocab_create": (i32, [vp, i32, vp, vp, vp, vp, i32, C.POINTER(vp)]),
        "orbgpu_vocab_destroy": (None, [vp]),
        "orbgpu_bow_transform": (i32, [vp, vp, vp, i32, i32, vp, vp]),
        "orbgpu_bow_transform_device": (i32, [vp, vp, vp, i32, i32, vp, vp]),
        "orbgpu_bow_score": (i32, [vp, C.POINTER(BowScoreArgs), vp, vp]),
        "orbgpu_bow_score_device": (i32, [vp, C.POINTER(BowScoreArgs), vp, vp]),
        "orbgpu_bowdb_create": (i32, [vp, i32, i32, C.POINTER(vp)]),
        "orbgpu_bowdb_destroy": (None, [vp]),
        "orbgpu_bowdb_append": (i32, [vp, u64, vp, vp, vp, i32]),
        "orbgpu_bowdb_erase": (i32, [vp, u64]),
        "orbgpu_bowdb_size": (i32, [vp, C.POINTER(i32)]),
        "orbgpu_bowdb_query": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, C.POINTER(i32)]),
        "orbgpu_rotation_filter": (i32, [vp, vp, i32, vp, i32, vp, C.POINTER(i32)]),
        "orbgpu_stage_times": (i32, [vp, vp]),
        "orbgpu_set_stage_timing": (i32, [vp, i32]),
        "orbgpu_popcount_peak": (i32, [vp, C.POINTER(dbl)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, handle=None):
    """Raise OrbGpuError for a non-zero status of a C-ABI call (message from orbgpu_last_error)."""
    if rc != OK:
        raise OrbGpuError(rc, load().orbgpu_last_error(handle).decode())


def ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))
