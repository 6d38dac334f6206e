"""GPU parity of the matching path (through the C ABI) against the CPU oracle: brute-force SNN, windowed
(AreaTo / ProjectionSearch) SNN, uniqueness, rotation histogram.  Match sets must be identical."""
import numpy as np
import pytest

from orb_slam3_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fe():
    from orb_slam3_amd.features import ORBFeatureExtractor
    e = ORBFeatureExtractor(640, 480, 1.2, 8, 20, 7, max_features=1000)
    yield e
    e.close()


def _as_dict(idx):
    return {i: int(j) for i, j in enumerate(idx) if j >= 0}


@pytest.mark.parametrize("n_to,n_from", [(1, 1), (80, 60), (1000, 1000), (3000, 257), (65, 5000)])
@pytest.mark.parametrize("th,ratio", [(50, 0.9), (100, 0.8), (50, 1.0), (100, 0.6)])
def test_bruteforce_parity(oracle, fe, n_to, n_from, th, ratio):
    from orb_slam3_amd.features import SNNMatcher
    frm = synth.descriptors(100 + n_from, n_from)
    to = synth.descriptors(200 + n_to, n_to, planted_from=frm, planted_frac=0.6, max_flips=70)
    if n_to > 10 and n_from > 12:
        to[5] = to[4]        # equal claims on one 'from': earliest 'to' keeps it
        frm[11] = frm[10]    # best == second: ratio test fails
    omf, omd = oracle.snn_match(to, frm, th, ratio)
    m = SNNMatcher(ratio, th, fe)
    gidx, gdist = m.MatchRaw(to, frm)
    assert np.array_equal(gidx, omf)
    sel = omf >= 0
    assert np.array_equal(gdist[sel], omd[sel])
    assert m.MatchBruteForce(to, frm) == _as_dict(omf)


def test_bruteforce_edge_cases(oracle, fe):
    from orb_slam3_amd.features import SNNMatcher
    m = SNNMatcher(0.9, 50, fe)
    frm = synth.descriptors(1, 10)
    assert m.MatchBruteForce(np.zeros((0, 32), np.uint8), frm) == {}
    assert m.MatchBruteForce(frm, np.zeros((0, 32), np.uint8)) == {}
    # identical sets: everything matches itself at distance 0 (0 < 0.9 * second unless second == 0)
    got = m.MatchBruteForce(frm, frm)
    assert got == {i: i for i in range(10)}
    # all-equal descriptors: best == second == 0 -> nothing passes
    same = np.repeat(frm[:1], 7, 0)
    assert m.MatchBruteForce(same, same) == {}
    # a single candidate passes the ratio test against UINT_MAX
    assert m.MatchBruteForce(frm[:3], frm[:1]) == {0: 0}


def _random_features(rng, n, w=640, h=480):
    from orb_slam3_amd.features import Features
    from orb_slam3_amd._lib import KP_DTYPE
    f = Features(w, h)
    k = np.zeros(n, KP_DTYPE)
    lv = rng.integers(0, 4, n)
    sc = np.float32(1.2) ** lv
    k["x"] = rng.integers(19, w - 19, n) * sc.astype(np.float64)
    k["y"] = rng.integers(19, h - 19, n) * sc.astype(np.float64)
    k["level"] = lv
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    f.keypoints = k
    f.descriptors = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    return f


def _to_oracle_kp(oracle, k):
    o = np.zeros(len(k), oracle.KP_DTYPE)
    for n in ("x", "y", "level"):
        o[n] = k[n]
    o["angle"] = k["angle"].astype(np.float64)
    o["size"] = k["size"]
    return o


@pytest.mark.parametrize("n,window,seed", [(300, 30, 1), (1000, 100, 2), (1000, 15, 3)])
def test_area_parity(oracle, fe, n, window, seed):
    from orb_slam3_amd.features import SNNMatcher
    rng = np.random.default_rng(seed)
    ff = _random_features(rng, n)
    ft = _random_features(rng, n)
    # make 'to' a jittered copy of 'from' so matches exist
    ft.keypoints["x"] = ff.keypoints["x"] + rng.integers(-6, 7, n)
    ft.keypoints["y"] = ff.keypoints["y"] + rng.integers(-6, 7, n)
    ft.keypoints["level"] = ff.keypoints["level"]
    ft.descriptors = ff.descriptors.copy()
    flips = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    ft.descriptors[:, :3] ^= flips
    ft.keypoints["level"][0] = 2  # id 0 is visited whatever its level
    omf, omd = oracle.match_area(_to_oracle_kp(oracle, ft.keypoints), ft.descriptors,
                                 _to_oracle_kp(oracle, ff.keypoints), ff.descriptors, window, 50, 0.9)
    got = SNNMatcher(0.9, 50, fe).MatchArea(ft, ff, window)
    assert got == _as_dict(omf) and len(got) > 10


@pytest.mark.parametrize("radius", [15.0, 30.0, 7.9])
def test_projection_parity(oracle, fe, radius):
    from orb_slam3_amd.features import SNNMatcher
    rng = np.random.default_rng(11)
    n = 800
    ff = _random_features(rng, n)
    vis = []
    sel = rng.choice(n, 400, replace=False)
    for j in sel:
        d = ff.descriptors[j].copy()
        d[:2] ^= rng.integers(0, 256, 2).astype(np.uint8)
        lvl = int(ff.keypoints["level"][j]) + int(rng.integers(-1, 2))
        vis.append({"descriptor": d, "position": (ff.keypoints["x"][j] + rng.uniform(-4, 4), ff.keypoints["y"][j] + rng.uniform(-4, 4)),
                    "window_size": radius * float(np.float32(1.2) ** max(lvl, 0)), "level": max(lvl, 0)})
    got = SNNMatcher(0.9, 100, fe).MatchProjection(vis, ff)
    desc = np.stack([v["descriptor"] for v in vis])
    xy = np.array([v["position"] for v in vis])
    win = np.array([int(v["window_size"]) for v in vis], np.uint64)
    lv = np.array([v["level"] for v in vis], np.int32)
    kf = ff.keypoints
    omf, _ = oracle.snn_match(desc, ff.descriptors, 100, 0.9, to_xy=xy, to_window=win, to_min_level=np.maximum(lv - 1, 0),
                              to_max_level=lv + 1, from_xy=np.stack([kf["x"], kf["y"]], 1), from_level=kf["level"])
    assert got == _as_dict(omf) and len(got) > 50


def test_rotation_filter_parity(oracle, fe):
    from orb_slam3_amd.features import OrientationValidator
    rng = np.random.default_rng(13)
    for n, noise in ((300, 12.0), (50, 40.0), (1000, 3.0)):
        ff = _random_features(rng, n)
        ft = _random_features(rng, n)
        ft.keypoints["angle"] = ((ff.keypoints["angle"] + rng.normal(20, noise, n)) % 360).astype(np.float32)
        ft.keypoints["angle"][: n // 8] = rng.uniform(0, 360, n // 8).astype(np.float32)
        matches = {int(i): int(i) for i in np.nonzero(rng.random(n) < 0.8)[0]}
        mf = np.full(n, -1, np.int32)
        for i, j in matches.items():
            mf[i] = j
        exp, end = oracle.orientation_validate(ft.keypoints["angle"].astype(np.float64),
                                               ff.keypoints["angle"].astype(np.float64), mf)
        got = dict(matches)
        nd = OrientationValidator(ft.keypoints, ff.keypoints, fe).Validate(got)
        assert nd == end and got == _as_dict(exp)
    assert OrientationValidator(ft.keypoints[:0], ff.keypoints[:0], fe).Validate({}) == 0


def test_extract_then_link_end_to_end(oracle):
    """extract two consecutive synthetic frames on the GPU, link them (window 100, 0.9 / 50 + rotation filter) and
    compare with the oracle run on the oracle's own extraction: MonocularFrame::ComputeMatchesForLinking."""
    from orb_slam3_amd.features import ORBFeatureExtractor, SNNMatcher, OrientationValidator
    frames = synth.sequence(1234, 2)
    fe = ORBFeatureExtractor(640, 480, 1.2, 8, 20, 7, max_batch=2, max_features=1000)
    f_from, f_to = fe.ExtractBatch([frames[0], frames[1]], 1000)
    p = oracle.params(640, 480)
    ok_from, od_from = oracle.extract(p, frames[0], 1000)
    ok_to, od_to = oracle.extract(p, frames[1], 1000)
    assert np.array_equal(f_to.descriptors, od_to) and np.array_equal(f_from.descriptors, od_from)
    got = SNNMatcher(0.9, 50, fe).MatchArea(f_to, f_from, 100)
    nd = OrientationValidator(f_to.keypoints, f_from.keypoints, fe).Validate(got)
    omf, _ = oracle.match_area(ok_to, od_to, ok_from, od_from, 100, 50, 0.9)
    omf, ond = oracle.orientation_validate(ok_to["angle"], ok_from["angle"], omf)
    assert got == _as_dict(omf) and nd == ond
    assert len(got) >= 100  # the reference needs >= 100 matches to link (monocular_frame.cpp:195)
    fe.close()


def _feature_vectors(rng, to, frm, n_nodes):
    """synthetic DBoW2::FeatureVector maps: node mostly decided by the descriptor, so planted pairs share a node"""
    node_of = lambda d: int(d[0]) % n_nodes
    fv_to, fv_from = {}, {}
    for i in range(len(to)):
        nd = node_of(to[i]) if rng.random() < 0.9 else int(rng.integers(0, n_nodes))
        fv_to.setdefault(2 * nd + (1 if rng.random() < 0.05 else 0), []).append(i)
    for j in range(len(frm)):
        nd = node_of(frm[j]) if rng.random() < 0.9 else int(rng.integers(0, n_nodes))
        fv_from.setdefault(2 * nd + (1 if rng.random() < 0.05 else 0), []).append(j)
    return fv_to, fv_from


@pytest.mark.parametrize("n_to,n_from,n_nodes", [(1, 1, 1), (300, 280, 7), (1000, 1000, 100), (2500, 900, 3), (64, 4000, 40)])
def test_bow_group_parity(oracle, fe, n_to, n_from, n_nodes):
    """orbgpu_match_groups (DBoW2Handler::FastMatch's node-restricted matcher) against the oracle for the three
    severities of dbo_w2_handler.cpp:37-51, followed by the rotation filter as in :68-71."""
    from orb_slam3_amd.features import SNNMatcher, OrientationValidator
    rng = np.random.default_rng(n_to + n_from)
    frm = synth.descriptors(300 + n_from, n_from)
    to = synth.descriptors(400 + n_to, n_to, planted_from=frm, planted_frac=0.6, max_flips=70)
    if n_to > 10 and n_from > 12:
        to[5] = to[4]
        frm[11] = frm[10]
    fv_to, fv_from = _feature_vectors(rng, to, frm, n_nodes)
    fv_to[2 * n_nodes + 10] = []                 # shared node with an empty 'to' list (bow_to_iterator.cpp:31-35)
    fv_from[2 * n_nodes + 10] = [0]
    fv_to[2 * n_nodes + 20] = [0] if 0 not in sum(fv_to.values(), []) else []   # node only on the 'to' side
    ang_to = rng.choice([5.0, 37.0, 181.0, 359.0], n_to).astype(np.float32)
    ang_from = rng.choice([5.0, 36.0, 180.0, 2.0], n_from).astype(np.float32)
    for th, ratio in ((50, 0.6), (50, 0.7), (100, 0.9)):
        omf, _ = oracle.match_bow(to, frm, fv_to, fv_from, th, ratio)
        got = SNNMatcher(ratio, th, fe).MatchBow(to, frm, fv_to, fv_from)
        assert got == _as_dict(omf)
        ov, _ = oracle.orientation_validate(ang_to, ang_from, omf)
        from orb_slam3_amd._lib import KP_DTYPE
        kt, kf = np.zeros(n_to, KP_DTYPE), np.zeros(n_from, KP_DTYPE)
        kt["angle"], kf["angle"] = ang_to, ang_from
        OrientationValidator(kt, kf, fe).Validate(got)
        assert got == _as_dict(ov)


def test_bow_group_errors(fe):
    from orb_slam3_amd import _lib
    import ctypes as C
    lib = _lib.load()
    d = np.zeros((4, 32), np.uint8)
    idx, dist = np.zeros(4, np.int32), np.zeros(4, np.uint32)
    off = np.array([0, 2], np.int32)
    ids = np.array([0, 9], np.uint32)      # 9 is outside [0, 4)
    a = _lib.GroupArgs(_lib.ptr(d), 4, _lib.ptr(d), 4, 1, _lib.ptr(off), _lib.ptr(ids), _lib.ptr(off), _lib.ptr(ids), 50, 0.9)
    assert lib.orbgpu_match_groups(fe.handle, C.byref(a), _lib.ptr(idx), _lib.ptr(dist)) == _lib.E_INVALID
    bad = np.array([0, 2, 1], np.int32)    # decreasing offsets
    ok_ids = np.array([0, 1], np.uint32)
    a = _lib.GroupArgs(_lib.ptr(d), 4, _lib.ptr(d), 4, 2, _lib.ptr(bad), _lib.ptr(ok_ids), _lib.ptr(bad), _lib.ptr(ok_ids), 50, 0.9)
    assert lib.orbgpu_match_groups(fe.handle, C.byref(a), _lib.ptr(idx), _lib.ptr(dist)) == _lib.E_INVALID
    a = _lib.GroupArgs(_lib.ptr(d), 4, _lib.ptr(d), 4, 0, None, None, None, None, 50, 0.9)   # no shared node: no matches
    assert lib.orbgpu_match_groups(fe.handle, C.byref(a), _lib.ptr(idx), _lib.ptr(dist)) == 0
    assert (idx == -1).all()


def test_distinctive_descriptor_parity(oracle, fe):
    """orbgpu_distinctive_descriptor (MapPoint::ComputeDistinctiveDescriptor, batched) against the oracle."""
    from orb_slam3_amd.features import ComputeDistinctiveDescriptors
    rng = np.random.default_rng(14)
    sizes = [0, 1, 2, 3, 4, 5, 8, 13, 33, 64, 65, 127, 200] + rng.integers(1, 40, 3000).tolist()
    groups = []
    for n in sizes:
        centre = rng.integers(0, 256, 32, dtype=np.uint8)
        g = np.repeat(centre[None], n, 0)
        flips = rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8) & rng.integers(0, 256, (n, 32), dtype=np.uint8)
        g = g ^ flips                      # observations of one point: noisy copies of a centre
        if n > 3 and n % 3 == 0:
            g[2] = g[0]                    # ties
        groups.append(g)
    off = np.cumsum([0] + [len(g) for g in groups]).astype(np.int32)
    want = oracle.distinctive_descriptor(np.concatenate(groups), off)
    got = ComputeDistinctiveDescriptors(fe, groups)
    assert np.array_equal(got, want)
    assert len(ComputeDistinctiveDescriptors(fe, [])) == 0
    assert ComputeDistinctiveDescriptors(fe, [np.zeros((0, 32), np.uint8)])[0] == -1


def test_keypoint_geometry_parity(oracle, fe):
    """orbgpu_keypoint_geometry (UndistortPoint + UnprojectAndUndistort per keypoint) against the oracle:
    bit-exact without a model and for Barrel5 (+, -, *, / only); FishEye goes through tan(): <= 4 ulp."""
    from orb_slam3_amd.features import Features, MonocularCamera
    from orb_slam3_amd._lib import KP_DTYPE, OrbGpuError
    from test_oracle import CAMERAS
    rng = np.random.default_rng(3)
    n = 5000
    f = Features(640, 480)
    f.keypoints = np.zeros(n, KP_DTYPE)
    f.keypoints["x"], f.keypoints["y"] = rng.uniform(0, 640, n), rng.uniform(0, 480, n)
    f.keypoints["x"][:3], f.keypoints["y"][:3] = (320.5, 0, 639), (240.25, 0, 479)
    okp = np.zeros(n, oracle.KP_DTYPE)
    okp["x"], okp["y"] = f.keypoints["x"], f.keypoints["y"]
    for model, k in CAMERAS:
        cam = MonocularCamera(517.3, 516.5, 320.5, 240.25, model, k)
        ok = cam.UndistortFeatures(fe, f)
        wund, wunp, wok = oracle.keypoint_geometry(okp, oracle.camera(517.3, 516.5, 320.5, 240.25, model, k))
        if model == 2:
            # a point whose last Newton step sits at the 7e-8 convergence limit may flip with a 1-ulp tan difference
            same = ok == wok
            assert same.mean() > 0.999
            # unprojected rays: a few ulp of tan(); pixels add c_x to f_x * u (cancellation near the image origin), so
            # their error is a few ulp of the image size, not of the value
            got, want = f.undistorted_and_unprojected_keypoints[same], wunp[same]
            assert (np.abs(got - want) <= 4 * np.spacing(np.abs(want))).all()
            assert (np.abs(f.undistorted_keypoints[same] - wund[same]) <= 1e-12).all()
        else:
            assert np.array_equal(ok, wok)
            assert np.array_equal(f.undistorted_keypoints, wund)
            assert np.array_equal(f.undistorted_and_unprojected_keypoints, wunp)
    with pytest.raises(OrbGpuError):
        MonocularCamera(500, 500, 320, 240, 1, (0,) * 8).UndistortFeatures(fe, f)     # BARREL8
    f0 = Features(640, 480)
    assert len(MonocularCamera(500, 500, 320, 240).UndistortFeatures(fe, f0)) == 0


def test_window_segments_parity(oracle, fe):
    """orbgpu_match_window_segments_device: many independent search-by-projection problems in one launch, ragged and
    empty segments included, padded layout (gaps between segments), against one oracle call per segment."""
    import ctypes as C
    import torch
    from orb_slam3_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(21)
    sizes = [(300, 280), (0, 50), (1, 1), (700, 0), (513, 900), (256, 257), (40, 1500)]
    cap_to, cap_from = 720, 1600                      # padded slots: segment s owns rows [s * cap, s * cap + count)
    S = len(sizes)
    to_desc = rng.integers(0, 256, (S * cap_to, 32), dtype=np.uint8)
    from_desc = rng.integers(0, 256, (S * cap_from, 32), dtype=np.uint8)
    to_xy = rng.uniform(0, 640, (S * cap_to, 2))
    from_xy = rng.uniform(0, 640, (S * cap_from, 2))
    to_lv = rng.integers(0, 8, S * cap_to).astype(np.int32)
    from_lv = rng.integers(0, 8, S * cap_from).astype(np.int32)
    win = (np.float64(24.0) * np.float64(1.2) ** to_lv).astype(np.uint64)        # truncated radius
    segs = np.zeros(S, _lib.SEGMENT_DTYPE)
    for s, (nt, nf) in enumerate(sizes):
        segs[s] = (s * cap_to, nt, s * cap_from, nf)
        k = min(nt, nf)
        if k:                                         # planted matches: a 'to' near a 'from' with a close descriptor
            j = rng.choice(nf, k, replace=False)
            i = rng.choice(nt, k, replace=False)
            to_desc[s * cap_to + i] = from_desc[s * cap_from + j]
            to_desc[s * cap_to + i, :2] ^= rng.integers(0, 256, (k, 2)).astype(np.uint8)
            to_xy[s * cap_to + i] = from_xy[s * cap_from + j] + rng.uniform(-5, 5, (k, 2))
            to_lv[s * cap_to + i] = np.clip(from_lv[s * cap_from + j] + rng.integers(-1, 2, k), 0, 7)
            if k > 8:                                 # contested claim and a best == second tie
                to_desc[s * cap_to + i[1]] = to_desc[s * cap_to + i[0]]
                to_xy[s * cap_to + i[1]] = to_xy[s * cap_to + i[0]]
                to_lv[s * cap_to + i[1]] = to_lv[s * cap_to + i[0]]
    win = (np.float64(24.0) * np.float64(1.2) ** to_lv).astype(np.uint64)
    lmin, lmax = np.maximum(to_lv - 1, 0).astype(np.int32), (to_lv + 1).astype(np.int32)
    dev = torch.device("cuda", 0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in dict(
        to_desc=to_desc, from_desc=from_desc, to_xy=to_xy, from_xy=from_xy, win=win.view(np.int64), lmin=lmin, lmax=lmax,
        from_lv=from_lv, segs=segs.view(np.int32).reshape(S, 4)).items()}
    d_idx = torch.full((S * cap_to,), 777, dtype=torch.int32, device=dev)
    d_dist = torch.full((S * cap_to,), 777, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for th, ratio in ((100, 0.9), (100, 0.8), (50, 1.0)):
        a = _lib.WindowArgs(t["to_desc"].data_ptr(), t["to_xy"].data_ptr(), t["win"].data_ptr(), t["lmin"].data_ptr(),
                            t["lmax"].data_ptr(), None, S * cap_to, t["from_desc"].data_ptr(), t["from_xy"].data_ptr(),
                            t["from_lv"].data_ptr(), S * cap_from, th, ratio)
        rc = lib.orbgpu_match_window_segments_device(fe.handle, C.byref(a), t["segs"].data_ptr(), S,
                                                     max(nt for nt, _ in sizes), d_idx.data_ptr(), d_dist.data_ptr())
        assert rc == 0, lib.orbgpu_last_error(fe.handle)
        assert lib.orbgpu_synchronize(fe.handle) == 0
        gi, gd = d_idx.cpu().numpy(), d_dist.cpu().numpy().view(np.uint32)
        total = 0
        for s, (nt, nf) in enumerate(sizes):
            ts, fs = slice(s * cap_to, s * cap_to + nt), slice(s * cap_from, s * cap_from + nf)
            omf, omd = oracle.snn_match(to_desc[ts], from_desc[fs], th, ratio, to_xy=to_xy[ts], to_window=win[ts],
                                        to_min_level=lmin[ts], to_max_level=lmax[ts], from_xy=from_xy[fs], from_level=from_lv[fs])
            assert np.array_equal(gi[ts], omf), (s, th, ratio)
            sel = omf >= 0
            assert np.array_equal(gd[ts][sel], omd[sel])
            assert (gi[s * cap_to + nt:(s + 1) * cap_to] == 777).all()       # rows outside the segment are not written
            total += int(sel.sum())
        assert total > 300


def test_grid_build_matches_assign_features_to_grid(oracle, fe):
    """orbgpu_grid_build_device against the oracle's Features::AssignFeaturesToGrid over the WORKING PosInGrid
    (features.cpp:127-161): cell boundaries, clamping of out-of-image points, ascending order inside a cell, ragged and
    empty ranges, one crowded cell."""
    import torch
    from orb_slam3_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    W, H = 1920.0, 1080.0
    counts = [5000, 0, 1, 777, 3000]
    cap = max(counts)
    xy = np.zeros((len(counts) * cap, 2))
    ranges = np.zeros((len(counts), 2), np.int32)
    for r, n in enumerate(counts):
        ranges[r] = (r * cap, n)
        pts = rng.uniform((-40, -40), (W + 40, H + 40), (n, 2))          # some outside the image: clamped to the border cells
        if n >= 777:
            pts[:64] = np.floor(pts[:64] / (W / 64)) * (W / 64)           # exactly on cell boundaries (x)
            pts[64:128, 1] = np.floor(pts[64:128, 1] / (H / 48)) * (H / 48)
            pts[128:140] = (W, H)                                         # the far corner: min(COLS - 1, ...)
        if n == 3000:
            pts[1000:2500] = rng.uniform((300, 300), (310, 310), (1500, 2))   # a crowded cell
        xy[r * cap:r * cap + n] = pts
    dev = torch.device("cuda", 0)
    d_xy, d_r = torch.from_numpy(xy).to(dev), torch.from_numpy(ranges).to(dev)
    d_cs = torch.full((len(counts), 3073), -5, dtype=torch.int32, device=dev)
    d_it = torch.full((len(counts), cap), -5, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for rep in range(2):
        rc = lib.orbgpu_grid_build_device(fe.handle, d_xy.data_ptr(), d_r.data_ptr(), len(counts), cap, W, H,
                                          d_cs.data_ptr(), d_it.data_ptr())
        assert rc == 0, lib.orbgpu_last_error(fe.handle)
    assert lib.orbgpu_synchronize(fe.handle) == 0
    cs, it = d_cs.cpu().numpy(), d_it.cpu().numpy()
    for r, n in enumerate(counts):
        ocs, oit = oracle.assign_features_to_grid(xy[r * cap:r * cap + n], W, H)
        assert np.array_equal(cs[r], ocs), r
        assert np.array_equal(it[r, :n], oit), r
        assert (it[r, n:] == -5).all()
    assert np.diff(cs[4]).max() >= 1500


@pytest.mark.parametrize("W,H,n_kp,radius", [(640, 480, 1000, 24.0), (1920, 1080, 5000, 15.0), (1920, 1080, 5000, 100.0),
                                             (3840, 2160, 3000, 30.0)])
def test_grid_window_segments_parity(oracle, fe, W, H, n_kp, radius):
    """orbgpu_match_window_grid_segments_device = orbgpu_match_window_segments_device = the oracle, on ragged segments
    with planted matches, ties, contested claims, inactive rows, keypoints outside the image and windows that reach
    over the border."""
    import ctypes as C
    import torch
    from orb_slam3_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(int(W + n_kp + radius))
    sizes = [(n_kp, n_kp), (0, 50), (1, 1), (n_kp // 2, 0), (n_kp // 3, n_kp), (257, n_kp - 7)]
    cap = n_kp
    S = len(sizes)
    to_desc = rng.integers(0, 256, (S * cap, 32), dtype=np.uint8)
    from_desc = rng.integers(0, 256, (S * cap, 32), dtype=np.uint8)
    to_xy = rng.uniform((-20, -20), (W + 20, H + 20), (S * cap, 2))
    from_xy = rng.uniform((-20, -20), (W + 20, H + 20), (S * cap, 2))
    to_lv = rng.integers(0, 8, S * cap).astype(np.int32)
    from_lv = rng.integers(0, 8, S * cap).astype(np.int32)
    active = (rng.uniform(size=S * cap) < 0.9).astype(np.uint8)
    segs = np.zeros(S, _lib.SEGMENT_DTYPE)
    for s, (nt, nf) in enumerate(sizes):
        segs[s] = (s * cap, nt, s * cap, nf)
        k = min(nt, nf)
        if k:
            j = rng.choice(nf, k, replace=False)
            i = rng.choice(nt, k, replace=False)
            to_desc[s * cap + i] = from_desc[s * cap + j]
            to_desc[s * cap + i, :2] ^= rng.integers(0, 256, (k, 2)).astype(np.uint8)
            to_xy[s * cap + i] = from_xy[s * cap + j] + rng.uniform(-5, 5, (k, 2))
            to_lv[s * cap + i] = np.clip(from_lv[s * cap + j] + rng.integers(-1, 2, k), 0, 7)
            if k > 8:
                for q in (1, 2):                                      # contested claim, best == second tie
                    to_desc[s * cap + i[q]] = to_desc[s * cap + i[0]]
                    to_xy[s * cap + i[q]] = to_xy[s * cap + i[0]]
                    to_lv[s * cap + i[q]] = to_lv[s * cap + i[0]]
                from_desc[s * cap + j[3]] = from_desc[s * cap + j[4]]   # equal distances: the smaller index must win
                from_xy[s * cap + j[3]] = from_xy[s * cap + j[4]]
                from_lv[s * cap + j[3]] = from_lv[s * cap + j[4]]
    win = (np.float64(radius) * np.float64(1.2) ** to_lv).astype(np.uint64)
    lmin, lmax = np.maximum(to_lv - 1, 0).astype(np.int32), (to_lv + 1).astype(np.int32)
    dev = torch.device("cuda", 0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in dict(
        to_desc=to_desc, from_desc=from_desc, to_xy=to_xy, from_xy=from_xy, win=win.view(np.int64), lmin=lmin, lmax=lmax,
        from_lv=from_lv, active=active, segs=segs.view(np.int32).reshape(S, 4)).items()}
    out = [(torch.full((S * cap,), 777, dtype=torch.int32, device=dev), torch.full((S * cap,), 777, dtype=torch.int32, device=dev))
           for _ in range(2)]
    torch.cuda.synchronize()
    max_to, max_from = max(nt for nt, _ in sizes), max(nf for _, nf in sizes)
    for th, ratio in ((100, 0.9), (50, 1.0)):
        a = _lib.WindowArgs(t["to_desc"].data_ptr(), t["to_xy"].data_ptr(), t["win"].data_ptr(), t["lmin"].data_ptr(),
                            t["lmax"].data_ptr(), t["active"].data_ptr(), S * cap, t["from_desc"].data_ptr(),
                            t["from_xy"].data_ptr(), t["from_lv"].data_ptr(), S * cap, th, ratio)
        rc = lib.orbgpu_match_window_segments_device(fe.handle, C.byref(a), t["segs"].data_ptr(), S, max_to,
                                                     out[0][0].data_ptr(), out[0][1].data_ptr())
        assert rc == 0, lib.orbgpu_last_error(fe.handle)
        rc = lib.orbgpu_match_window_grid_segments_device(fe.handle, C.byref(a), t["segs"].data_ptr(), S, max_to, max_from,
                                                          float(W), float(H), out[1][0].data_ptr(), out[1][1].data_ptr())
        assert rc == 0, lib.orbgpu_last_error(fe.handle)
        assert lib.orbgpu_synchronize(fe.handle) == 0
        pi, pd = out[0][0].cpu().numpy(), out[0][1].cpu().numpy()
        gi, gd = out[1][0].cpu().numpy(), out[1][1].cpu().numpy()
        assert np.array_equal(gi, pi) and np.array_equal(gd, pd), (th, ratio)
        total = 0
        for s, (nt, nf) in enumerate(sizes[:3] + sizes[5:]):          # the oracle on the small segments and one large one
            s = s if s < 3 else 5
            ts, fs = slice(s * cap, s * cap + nt), slice(s * cap, s * cap + nf)
            omf, omd = oracle.snn_match(to_desc[ts], from_desc[fs], th, ratio, to_xy=to_xy[ts], to_window=win[ts],
                                        to_min_level=lmin[ts], to_max_level=lmax[ts], to_active=active[ts],
                                        from_xy=from_xy[fs], from_level=from_lv[fs])
            assert np.array_equal(gi[ts], omf), (s, th, ratio)
            sel = omf >= 0
            assert np.array_equal(gd.view(np.uint32)[ts][sel], omd[sel])
            total += int(sel.sum())
        assert total > 100


def test_grid_segments_flags_an_understated_maximum(fe):
    """max_from_count / max_to_count smaller than a segment: the library stays inside that segment's scratch rows and
    orbgpu_synchronize reports ORBGPU_E_CAPACITY (the flag is cleared by that report)."""
    import ctypes as C
    import torch
    from orb_slam3_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(2)
    n = 500
    dev = torch.device("cuda", 0)
    desc = torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).to(dev)
    xy = torch.from_numpy(rng.uniform(0, 600, (n, 2))).to(dev)
    win = torch.full((n,), 20, dtype=torch.int64, device=dev)
    lv = torch.zeros(n, dtype=torch.int32, device=dev)
    segs = torch.tensor([[0, n, 0, n]], dtype=torch.int32, device=dev)
    oi = torch.zeros(n, dtype=torch.int32, device=dev)
    od = torch.zeros(n, dtype=torch.int32, device=dev)
    a = _lib.WindowArgs(desc.data_ptr(), xy.data_ptr(), win.data_ptr(), lv.data_ptr(), lv.data_ptr(), None, n,
                        desc.data_ptr(), xy.data_ptr(), lv.data_ptr(), n, 100, 0.9)
    torch.cuda.synchronize()
    for max_to, max_from in ((n, n - 100), (n - 100, n)):
        assert lib.orbgpu_match_window_grid_segments_device(fe.handle, C.byref(a), segs.data_ptr(), 1, max_to, max_from,
                                                            640.0, 480.0, oi.data_ptr(), od.data_ptr()) == 0
        assert lib.orbgpu_synchronize(fe.handle) == _lib.E_CAPACITY
        assert b"max_" in lib.orbgpu_last_error(fe.handle)
        assert lib.orbgpu_synchronize(fe.handle) == 0
    assert lib.orbgpu_match_window_grid_segments_device(fe.handle, C.byref(a), segs.data_ptr(), 1, n, n, 640.0, 480.0,
                                                        oi.data_ptr(), od.data_ptr()) == 0
    assert lib.orbgpu_synchronize(fe.handle) == 0
    assert (oi.cpu().numpy() == np.arange(n)).all()              # every row finds itself


@pytest.mark.parametrize("n_to,n_from,grid_min", [(300, 280, "1"), (64, 256, "1"), (2500, 2600, None), (1000, 5000, None)])
def test_single_window_search_over_the_grid(oracle, n_to, n_from, grid_min):
    """orbgpu_match_window (host pointers, one search: the call SNNMatcher<ProjectionSearchIterator> makes) takes the search
    grid from 2048 'from' rows on (ORBGPU_TUNE=window_grid_min=.. lowers that for the small cases here): same results as
    the oracle and as a context that never uses the grid (ORBGPU_TUNE=window_scan=1)."""
    import os
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    rng = np.random.default_rng(n_to + n_from)
    W, H = 1280, 720
    from_desc = rng.integers(0, 256, (n_from, 32), dtype=np.uint8)
    to_desc = rng.integers(0, 256, (n_to, 32), dtype=np.uint8)
    from_xy = rng.uniform((-10, -10), (W + 10, H + 10), (n_from, 2))
    to_xy = rng.uniform((-10, -10), (W + 10, H + 10), (n_to, 2))
    k = min(n_to, n_from) // 2
    i, j = rng.choice(n_to, k, replace=False), rng.choice(n_from, k, replace=False)
    to_desc[i] = from_desc[j]
    to_desc[i, :2] ^= rng.integers(0, 256, (k, 2)).astype(np.uint8)
    to_xy[i] = from_xy[j] + rng.uniform(-5, 5, (k, 2))
    to_lv = rng.integers(0, 8, n_to).astype(np.int32)
    from_lv = rng.integers(0, 8, n_from).astype(np.int32)
    to_lv[i] = np.clip(from_lv[j] + rng.integers(-1, 2, k), 0, 7)
    to_desc[i[1]], to_xy[i[1]], to_lv[i[1]] = to_desc[i[0]], to_xy[i[0]], to_lv[i[0]]      # contested claim
    win = (np.float64(15.0) * np.float64(1.2) ** to_lv).astype(np.uint64)
    lmin, lmax = np.maximum(to_lv - 1, 0).astype(np.int32), (to_lv + 1).astype(np.int32)
    active = (rng.uniform(size=n_to) < 0.9).astype(np.uint8)
    omf, omd = oracle.snn_match(to_desc, from_desc, 100, 0.9, to_xy=to_xy, to_window=win, to_min_level=lmin,
                                to_max_level=lmax, to_active=active, from_xy=from_xy, from_level=from_lv)
    assert (omf >= 0).sum() > k // 3
    for env in ({"ORBGPU_TUNE": "window_grid_min=" + grid_min} if grid_min else {}, {"ORBGPU_TUNE": "window_scan=1"}):
        os.environ.update(env)
        try:
            ctx = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_features=1000)
        finally:
            for name in env:
                os.environ.pop(name, None)
        a = _lib.WindowArgs(_lib.ptr(to_desc), _lib.ptr(to_xy), _lib.ptr(win), _lib.ptr(lmin), _lib.ptr(lmax),
                            _lib.ptr(active), n_to, _lib.ptr(from_desc), _lib.ptr(from_xy), _lib.ptr(from_lv), n_from, 100, 0.9)
        gi, gd = np.zeros(n_to, np.int32), np.zeros(n_to, np.uint32)
        import ctypes as C
        for rep in range(2):
            assert lib.orbgpu_match_window(ctx.handle, C.byref(a), _lib.ptr(gi), _lib.ptr(gd)) == 0, \
                lib.orbgpu_last_error(ctx.handle)
        assert np.array_equal(gi, omf), env
        assert np.array_equal(gd[omf >= 0], omd[omf >= 0]), env
        ctx.close()


def test_grid_build_crowded_bins(oracle, fe):
    """Bins of thousands of rows (above the rank-counting limit of 4096): keypoints of a 4K image binned by a grid laid
    over 640 x 480 -- almost everything clamps into the last row / column of cells -- and a tight cluster.  The grid
    build switches to its linear stable placement; cell contents stay in ascending row order (push_back order)."""
    import torch
    from orb_slam3_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(9)
    counts = [20000, 9000, 300]
    cap = max(counts)
    xy = np.zeros((len(counts) * cap, 2))
    ranges = np.zeros((len(counts), 2), np.int32)
    for r, n in enumerate(counts):
        ranges[r] = (r * cap, n)
        if r == 0:
            pts = rng.uniform((0, 0), (3840, 2160), (n, 2))                 # 4K keypoints, 640 x 480 grid
        elif r == 1:
            pts = rng.uniform((100, 100), (640, 480), (n, 2))
            pts[1000:7000] = rng.uniform((300, 300), (309.9, 309.9), (6000, 2))   # 6000 rows in one cell
        else:
            pts = rng.uniform((0, 0), (640, 480), (n, 2))                   # an ordinary range beside the crowded ones
        xy[r * cap:r * cap + n] = pts
    dev = torch.device("cuda", 0)
    d_xy, d_r = torch.from_numpy(xy).to(dev), torch.from_numpy(ranges).to(dev)
    d_cs = torch.full((len(counts), 3073), -5, dtype=torch.int32, device=dev)
    d_it = torch.full((len(counts), cap), -5, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    rc = lib.orbgpu_grid_build_device(fe.handle, d_xy.data_ptr(), d_r.data_ptr(), len(counts), cap, 640.0, 480.0,
                                      d_cs.data_ptr(), d_it.data_ptr())
    assert rc == 0, lib.orbgpu_last_error(fe.handle)
    assert lib.orbgpu_synchronize(fe.handle) == 0
    cs, it = d_cs.cpu().numpy(), d_it.cpu().numpy()
    for r, n in enumerate(counts):
        ocs, oit = oracle.assign_features_to_grid(xy[r * cap:r * cap + n], 640.0, 480.0)
        same = np.array_equal(cs[r], ocs) and np.array_equal(it[r, :n], oit)
        assert same, r
    assert np.diff(cs[0]).max() > 4096 and np.diff(cs[1]).max() >= 6000
