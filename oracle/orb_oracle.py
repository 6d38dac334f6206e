"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (orb_slam3_amd) never does.  See orb_oracle.h for the parity status.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "liborb_oracle.so"

MAX_LEVELS = 16


class Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("scale_factor", C.c_double), ("levels", C.c_int),
                ("fast_init", C.c_int), ("fast_min", C.c_int)]


class Geometry(C.Structure):
    _fields_ = [("levels", C.c_int)] + [(n, C.c_int * MAX_LEVELS) for n in
                                        ("w", "h", "budget", "ncols", "nrows", "wcell", "hcell", "nini", "kp_size")] + \
               [(n, C.c_double * MAX_LEVELS) for n in ("scale", "inv_scale", "sigma2")]


KP_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("angle", "<f8"), ("size", "<f8"), ("level", "<i4"), ("pad", "<i4")])
CAND_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4")])


class Dump(C.Structure):
    _fields_ = [("pyramid", C.c_void_p), ("blurred", C.c_void_p), ("score", C.c_void_p), ("cands", C.c_void_p),
                ("cands_cap", C.c_int), ("n_cands", C.c_int * MAX_LEVELS), ("n_kp", C.c_int * MAX_LEVELS),
                ("tie_ambiguous", C.c_int * MAX_LEVELS)]


class MatchArgs(C.Structure):
    _fields_ = [("to_desc", C.c_void_p), ("n_to", C.c_int), ("from_desc", C.c_void_p), ("n_from", C.c_int),
                ("to_xy", C.c_void_p), ("to_window", C.c_void_p), ("to_min_level", C.c_void_p),
                ("to_max_level", C.c_void_p), ("to_active", C.c_void_p), ("from_xy", C.c_void_p),
                ("from_level", C.c_void_p), ("threshold", C.c_uint), ("ratio", C.c_double)]


def build(force=False):
    if force or not _LIB.exists() or _LIB.stat().st_mtime < (_HERE / "orb_oracle.c").stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE)], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _LIB.exists():
            build()
        _lib = C.CDLL(str(_LIB))
        _lib.oo_fast_atan2.restype = C.c_float
        _lib.oo_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.oo_ic_angle.restype = C.c_float
        _lib.oo_hamming256.restype = C.c_uint
        _lib.oo_predict_scale.restype = C.c_uint
        _lib.oo_predict_scale.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def params(width, height, scale=1.2, levels=8, fast_init=20, fast_min=7):
    return Params(int(width), int(height), float(scale), int(levels), int(fast_init), int(fast_min))


def geometry(p, n_features):
    g = Geometry()
    rc = lib().oo_geometry_compute(C.byref(p), int(n_features), C.byref(g))
    if rc:
        raise ValueError(f"unsupported geometry rc={rc}")
    return g


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.empty((dh, dw), np.uint8)
    rc = lib().oo_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    if rc:
        raise ValueError(rc)
    return dst


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.empty_like(src)
    lib().oo_gaussian_blur7(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def fast_atan2(y, x):
    return float(lib().oo_fast_atan2(float(y), float(x)))


def ic_angle(img, cx, cy):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    return float(lib().oo_ic_angle(_p(img), C.c_int(img.strides[0]), C.c_int(cx), C.c_int(cy)))


def orb_descriptor(blurred, cx, cy, angle_deg):
    blurred = np.ascontiguousarray(blurred, dtype=np.uint8)
    d = np.empty(32, np.uint8)
    lib().oo_orb_descriptor(_p(blurred), C.c_int(blurred.strides[0]), C.c_int(cx), C.c_int(cy),
                            C.c_double(angle_deg), _p(d))
    return d


def set_trig_variant(variant):
    """0: (float)cos((double)a) (default), 1: cosf(a) of the host libm -- see oo_descriptor_trig."""
    lib().oo_set_trig_variant(int(variant))


def descriptor_trig(angle_deg, variant):
    c, s = C.c_float(0), C.c_float(0)
    lib().oo_descriptor_trig(C.c_double(float(angle_deg)), int(variant), C.byref(c), C.byref(s))
    return c.value, s.value


def fast_score_map(img, min_thr=7):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty_like(img)
    lib().oo_fast_score_map(_p(img), img.shape[1], img.shape[0], img.strides[0], int(min_thr), _p(out), out.strides[0])
    return out


def fast_literal(roi, thr, nonmax=True):
    roi = np.ascontiguousarray(roi, dtype=np.uint8)
    cap = roi.size
    ox = np.empty(cap, np.int32)
    oy = np.empty(cap, np.int32)
    osc = np.empty(cap, np.int32)
    n = lib().oo_fast_literal(_p(roi), C.c_int(roi.strides[0]), roi.shape[1], roi.shape[0], int(thr), int(nonmax),
                              cap, _p(ox), _p(oy), _p(osc))
    return ox[:n].copy(), oy[:n].copy(), osc[:n].copy()


def detect_level(img, fast_init=20, fast_min=7, literal=False):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    cap = img.size // 4 + 64
    out = np.empty(cap, CAND_DTYPE)
    n = lib().oo_detect_level(_p(img), img.shape[1], img.shape[0], img.strides[0], int(fast_init), int(fast_min),
                              int(literal), _p(out), cap)
    if n < 0:
        raise ValueError(n)
    return out[:n].copy()


def distribute_octtree(cands, w, h, n_features):
    """cands: CAND_DTYPE array for a level of size w x h. Returns (x, y, resp, tie_flags) relative to minBorder."""
    cands = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
    cap = len(cands) + 8
    ox = np.empty(cap, np.float32)
    oy = np.empty(cap, np.float32)
    orr = np.empty(cap, np.float32)
    tie = C.c_int(0)
    n = lib().oo_distribute_octtree(_p(cands), len(cands), 16, w - 16, 16, h - 16, int(n_features), _p(ox), _p(oy),
                                    _p(orr), cap, C.byref(tie))
    if n < 0:
        raise ValueError(n)
    return ox[:n].copy(), oy[:n].copy(), orr[:n].copy(), tie.value


def extract(p, img, n_features, dump=False):
    """Returns (kps[KP_DTYPE], desc[N,32]) and, with dump=True, a dict of intermediates."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.size == 0:
        return -1
    g = geometry(p, n_features)
    cap = n_features + 4 * g.levels + 64
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    d = None
    extra = None
    if dump:
        tot = sum(g.w[l] * g.h[l] for l in range(g.levels))
        extra = {"pyramid": np.zeros(tot, np.uint8), "blurred": np.zeros(tot, np.uint8),
                 "score": np.zeros(tot, np.uint8), "cands": np.zeros(tot // 4 + 1024, CAND_DTYPE)}
        d = Dump(_p(extra["pyramid"]), _p(extra["blurred"]), _p(extra["score"]), _p(extra["cands"]),
                 len(extra["cands"]))
    n = lib().oo_extract(C.byref(p), _p(img), C.c_int(img.strides[0]), int(n_features), _p(kps), _p(desc), cap,
                         C.byref(d) if d is not None else None)
    if n < 0:
        raise ValueError(f"oo_extract rc={n}")
    if not dump:
        return kps[:n].copy(), desc[:n].copy()
    extra["n_cands"] = [d.n_cands[l] for l in range(g.levels)]
    extra["n_kp"] = [d.n_kp[l] for l in range(g.levels)]
    extra["tie"] = [d.tie_ambiguous[l] for l in range(g.levels)]
    extra["geometry"] = g
    return kps[:n].copy(), desc[:n].copy(), extra


def extract_literal(p, img, n_features):
    """Same outputs as extract(); FAST runs as literal per-cell cv::FAST. This is what bench.py times."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    g = geometry(p, n_features)
    cap = n_features + 4 * g.levels + 64
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = lib().oo_extract_literal(C.byref(p), _p(img), C.c_int(img.strides[0]), int(n_features), _p(kps), _p(desc), cap)
    if n < 0:
        raise ValueError(f"oo_extract_literal rc={n}")
    return kps[:n].copy(), desc[:n].copy()


def hamming256(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return int(lib().oo_hamming256(_p(a), _p(b)))


def list_features_in_area(kps, px, py, window, min_level, max_level):
    kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
    out = np.empty(len(kps) + 1, np.int32)
    n = lib().oo_list_features_in_area(_p(kps), len(kps), C.c_double(px), C.c_double(py), C.c_size_t(int(window)),
                                       int(min_level), int(max_level), _p(out))
    return out[:n].copy()


GRID_COLS, GRID_ROWS = 64, 48


def assign_features_to_grid(xy, width, height):
    """Features::AssignFeaturesToGrid over the working PosInGrid: (cell_start[3073], items[n]), cell = ix * 48 + iy."""
    xy = np.ascontiguousarray(xy, np.float64).reshape(-1, 2)
    cs = np.zeros(GRID_COLS * GRID_ROWS + 1, np.int32)
    items = np.zeros(max(len(xy), 1), np.int32)
    lib().oo_assign_features_to_grid(_p(xy), len(xy), C.c_double(width), C.c_double(height), _p(cs), _p(items))
    return cs, items[:len(xy)].copy()


def list_features_in_area_grid(xy, level, cell_start, items, width, height, px, py, window, min_level, max_level):
    xy = np.ascontiguousarray(xy, np.float64).reshape(-1, 2)
    level = np.ascontiguousarray(level, np.int32)
    cell_start = np.ascontiguousarray(cell_start, np.int32)
    items = np.ascontiguousarray(items, np.int32)
    out = np.empty(len(xy) + 1, np.int32)
    n = lib().oo_list_features_in_area_grid(_p(xy), _p(level), _p(cell_start), _p(items), C.c_double(width),
                                            C.c_double(height), C.c_double(px), C.c_double(py), C.c_size_t(int(window)),
                                            int(min_level), int(max_level), _p(out))
    return out[:n].copy()


def snn_match(to_desc, from_desc, threshold, ratio, to_xy=None, to_window=None, to_min_level=None,
              to_max_level=None, to_active=None, from_xy=None, from_level=None):
    to_desc = np.ascontiguousarray(to_desc, np.uint8).reshape(-1, 32)
    from_desc = np.ascontiguousarray(from_desc, np.uint8).reshape(-1, 32)
    keep = [to_desc, from_desc]

    def prep(a, dt):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a

    to_xy = prep(to_xy, np.float64)
    to_window = prep(to_window, np.uint64)
    to_min_level = prep(to_min_level, np.int32)
    to_max_level = prep(to_max_level, np.int32)
    to_active = prep(to_active, np.uint8)
    from_xy = prep(from_xy, np.float64)
    from_level = prep(from_level, np.int32)
    a = MatchArgs(_p(to_desc), len(to_desc), _p(from_desc), len(from_desc), _p(to_xy), _p(to_window),
                  _p(to_min_level), _p(to_max_level), _p(to_active), _p(from_xy), _p(from_level), int(threshold),
                  float(ratio))
    mf = np.empty(max(len(to_desc), 1), np.int32)
    md = np.zeros(max(len(to_desc), 1), np.uint32)
    lib().oo_snn_match(C.byref(a), _p(mf), _p(md))
    return mf[:len(to_desc)], md[:len(to_desc)]


def match_area(kp_to, desc_to, kp_from, desc_from, window=100, threshold=50, ratio=0.9):
    kp_to = np.ascontiguousarray(kp_to, dtype=KP_DTYPE)
    kp_from = np.ascontiguousarray(kp_from, dtype=KP_DTYPE)
    desc_to = np.ascontiguousarray(desc_to, np.uint8)
    desc_from = np.ascontiguousarray(desc_from, np.uint8)
    mf = np.empty(max(len(kp_to), 1), np.int32)
    md = np.zeros(max(len(kp_to), 1), np.uint32)
    lib().oo_match_area(_p(kp_to), _p(desc_to), len(kp_to), _p(kp_from), _p(desc_from), len(kp_from),
                        C.c_size_t(int(window)), C.c_uint(int(threshold)), C.c_double(ratio), _p(mf), _p(md))
    return mf[:len(kp_to)], md[:len(kp_to)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("model", C.c_int32),
                ("reserved", C.c_int32), ("k", C.c_double * 8)]


def camera(fx, fy, cx, cy, model=-1, k=()):
    c = Camera(fx, fy, cx, cy, model, 0)
    for i, v in enumerate(k):
        c.k[i] = v
    return c


def keypoint_geometry(kp, cam):
    kp = np.ascontiguousarray(kp, dtype=KP_DTYPE)
    n = len(kp)
    und = np.zeros((max(n, 1), 2), np.float64)
    unp = np.zeros((max(n, 1), 3), np.float64)
    ok = np.zeros(max(n, 1), np.uint8)
    lib().oo_keypoint_geometry(_p(kp), n, C.byref(cam), _p(und), _p(unp), _p(ok))
    return und[:n], unp[:n], ok[:n]


def distinctive_descriptor(desc, offsets):
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    best = np.zeros(max(len(offsets) - 1, 1), np.int32)
    lib().oo_distinctive_descriptor(_p(desc), _p(offsets), len(offsets) - 1, _p(best))
    return best[:len(offsets) - 1]


def bow_transform(desc, child_off, children, node_desc, word_id, L, levelsup=4):
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    child_off = np.ascontiguousarray(child_off, np.int32)
    children = np.ascontiguousarray(children, np.uint32)
    node_desc = np.ascontiguousarray(node_desc, np.uint8).reshape(-1, 32)
    word_id = np.ascontiguousarray(word_id, np.uint32)
    ow = np.zeros(max(len(desc), 1), np.uint32)
    on = np.zeros(max(len(desc), 1), np.uint32)
    lib().oo_bow_transform(_p(desc), len(desc), _p(child_off), _p(children), _p(node_desc), _p(word_id), int(L),
                           int(levelsup), _p(ow), _p(on))
    return ow[:len(desc)], on[:len(desc)]


def match_bow(desc_to, desc_from, fv_to, fv_from, threshold=50, ratio=0.9):
    """fv_to / fv_from: DBoW2::FeatureVector as {node id: [feature ids]} (any mapping; iterated in node order)."""
    desc_to = np.ascontiguousarray(desc_to, np.uint8).reshape(-1, 32)
    desc_from = np.ascontiguousarray(desc_from, np.uint8).reshape(-1, 32)

    def csr(fv):
        nodes = np.array(sorted(fv), np.uint32)
        off = np.zeros(len(nodes) + 1, np.int32)
        ids = []
        for k, nd in enumerate(nodes.tolist()):
            ids.extend(fv[nd])
            off[k + 1] = len(ids)
        return nodes, off, np.array(ids, np.uint32)

    tn, to_, ti = csr(fv_to)
    fn_, fo, fi = csr(fv_from)
    mf = np.empty(max(len(desc_to), 1), np.int32)
    md = np.zeros(max(len(desc_to), 1), np.uint32)
    n = lib().oo_match_bow(_p(desc_to), len(desc_to), _p(desc_from), len(desc_from), _p(tn), _p(to_), _p(ti), len(tn),
                           _p(fn_), _p(fo), _p(fi), len(fn_), C.c_uint(int(threshold)), C.c_double(ratio), _p(mf), _p(md))
    assert n == int((mf[:len(desc_to)] >= 0).sum())
    return mf[:len(desc_to)], md[:len(desc_to)]


def orientation_validate(angle_to, angle_from, match_from):
    angle_to = np.ascontiguousarray(angle_to, np.float64)
    angle_from = np.ascontiguousarray(angle_from, np.float64)
    m = np.array(match_from, dtype=np.int32, copy=True)
    nd = lib().oo_orientation_validate(_p(angle_to), _p(angle_from), _p(m), len(m))
    return m, nd


def predict_scale(distance, max_distance, scale_factor, levels):
    return int(lib().oo_predict_scale(distance, max_distance, scale_factor, levels))


def bow_score(q_words, q_values, kf_offsets, kf_words, kf_values, scoring=0, q_counts=None, kf_counts=None):
    """-> (shared word counts u64[n_kf], scores f64[n_kf]); see oo_bow_score."""
    q_words = np.ascontiguousarray(q_words, np.uint32)
    q_values = np.ascontiguousarray(q_values, np.float64)
    kf_offsets = np.ascontiguousarray(kf_offsets, np.int64)
    kf_words = np.ascontiguousarray(kf_words, np.uint32)
    kf_values = np.ascontiguousarray(kf_values, np.float64)
    qc = None if q_counts is None else np.ascontiguousarray(q_counts, np.uint32)
    kc = None if kf_counts is None else np.ascontiguousarray(kf_counts, np.uint32)
    n_kf = len(kf_offsets) - 1
    shared = np.zeros(max(n_kf, 1), np.uint64)
    score = np.zeros(max(n_kf, 1), np.float64)
    lib().oo_bow_score(_p(q_words), _p(q_values), _p(qc) if qc is not None else None, len(q_words), _p(kf_offsets),
                       _p(kf_words), _p(kf_values), _p(kc) if kc is not None else None, n_kf, int(scoring), _p(shared),
                       _p(score))
    return shared[:n_kf], score[:n_kf]
