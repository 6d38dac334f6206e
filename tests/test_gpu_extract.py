"""GPU parity of the extraction path (through the C ABI) against the CPU oracle, stage by stage."""
import numpy as np
import pytest

from orb_slam3_amd import synth

pytestmark = pytest.mark.gpu


def _levels(flat, g):
    out, off = [], 0
    for l in range(g.levels):
        n = g.w[l] * g.h[l]
        out.append(flat[off:off + n].reshape(g.h[l], g.w[l]))
        off += n
    return out


def _kp_equal(gk, ok):
    """keypoint coordinates / level / size bit-exact, angle bit-exact as float32"""
    assert len(gk) == len(ok)
    assert np.array_equal(gk["x"], ok["x"]) and np.array_equal(gk["y"], ok["y"])
    assert np.array_equal(gk["level"], ok["level"])
    assert np.array_equal(gk["size"].astype(np.float64), ok["size"])
    assert np.array_equal(gk["angle"], ok["angle"].astype(np.float32))


@pytest.mark.parametrize("w,h,nfeat,seed", [(640, 480, 1000, 1234), (320, 240, 500, 21), (512, 512, 1500, 3),
                                            (421, 307, 700, 5)])
def test_extract_stage_parity(oracle, w, h, nfeat, seed):
    from orb_slam3_amd.features import ORBFeatureExtractor, Features
    img = synth.frame(seed, h, w)
    p = oracle.params(w, h)
    okp, odesc, ex = oracle.extract(p, img, nfeat, dump=True)
    fe = ORBFeatureExtractor(w, h, 1.2, 8, 20, 7, max_features=nfeat)
    ft = Features(w, h)
    assert fe.Extract(img, ft, nfeat) == 0
    d = fe.debug_dump()
    g = ex["geometry"]
    for l, (a, b) in enumerate(zip(_levels(d["pyramid"], g), _levels(ex["pyramid"], g))):
        assert np.array_equal(a, b), f"pyramid level {l}"
    # the score map is an implementation intermediate: the cells only ever read [19, w-20] x [19, h-20] (their
    # detection rectangles; neighbours outside a rectangle count as 0), so that is what the kernel fills
    for l, (a, b) in enumerate(zip(_levels(d["score"], g), _levels(ex["score"], g))):
        assert np.array_equal(a[19:-19, 19:-19], b[19:-19, 19:-19]), f"FAST score level {l}"
        assert not a[:19].any() and not a[-19:].any() and not a[:, :19].any() and not a[:, -19:].any()
    assert d["n_cands"] == ex["n_cands"]
    n = sum(ex["n_cands"])
    oc = ex["cands"][:n]
    assert np.array_equal(d["cand_x"], oc["x"].astype(np.int32))
    assert np.array_equal(d["cand_y"], oc["y"].astype(np.int32))
    assert np.array_equal(d["cand_score"], oc["response"].astype(np.int32))
    for l, (a, b) in enumerate(zip(_levels(d["blurred"], g), _levels(ex["blurred"], g))):
        if ex["n_kp"][l]:
            assert np.array_equal(a, b), f"blur level {l}"
    _kp_equal(ft.keypoints, okp)
    assert np.array_equal(ft.descriptors, odesc)
    fe.close()


@pytest.mark.parametrize("w,h,nfeat,levels,seed,scene", [
    (1920, 1080, 5000, 8, 7, "rich"),    # two quadtree roots (nIni = 2), keys beyond the LDS buffers
    (640, 480, 30, 8, 8, "rich"),        # tiny budgets: the list overshoots / stops in the first sweeps
    (640, 480, 6000, 8, 9, "rich"),      # budget above the number of candidates: every node ends as a singleton
    (480, 640, 800, 6, 10, "rich"),      # portrait
    (800, 300, 1200, 5, 11, "rich"),     # wide: nIni = 3
    (1100, 200, 900, 3, 14, "rich"),     # very wide: nIni = 6, more roots than the block partition has classes
    (640, 480, 1000, 8, 12, "low"),      # low contrast: per-cell threshold fallback everywhere
    (640, 480, 1000, 8, 13, "noise"),    # white noise: very many candidates per cell
])
def test_extract_quadtree_stress(oracle, w, h, nfeat, levels, seed, scene):
    """keypoints / descriptors of the device quadtree path against the oracle on inputs that stress DistributeOctTree"""
    from orb_slam3_amd.features import ORBFeatureExtractor, Features
    img = synth.frame(seed, h, w)
    if scene == "low":
        img = (img.astype(np.int32) // 8 + 100).astype(np.uint8)
    elif scene == "noise":
        img = np.random.default_rng(seed).integers(0, 256, (h, w)).astype(np.uint8)
    p = oracle.params(w, h, levels=levels)
    okp, odesc = oracle.extract_literal(p, img, nfeat)
    fe = ORBFeatureExtractor(w, h, 1.2, levels, 20, 7, max_features=nfeat)
    ft = Features(w, h)
    assert fe.Extract(img, ft, nfeat) == 0
    _kp_equal(ft.keypoints, okp)
    assert np.array_equal(ft.descriptors, odesc)
    fe.close()


def _scene(kind, seed, h, w):
    rng = np.random.default_rng(seed)
    if kind == "checker":          # exact ties everywhere: equal scores, equal node sizes, equal responses
        yy, xx = np.mgrid[0:h, 0:w]
        return ((((yy // 9) + (xx // 11)) & 1) * 180 + 30).astype(np.uint8)
    if kind == "saturated":        # clipped blobs on black: rings full of 0 / 255
        img = np.zeros((h, w), np.uint8)
        for _ in range(300):
            y, x, r = int(rng.integers(0, h)), int(rng.integers(0, w)), int(rng.integers(2, 12))
            img[max(y - r, 0):y + r, max(x - r, 0):x + r] = 255
        return img
    if kind == "gradient":         # smooth ramp + sparse dots: most cells empty even at the low threshold
        yy, xx = np.mgrid[0:h, 0:w]
        img = ((yy * 255 // h + xx * 255 // w) // 2).astype(np.uint8)
        for _ in range(40):
            img[int(rng.integers(20, h - 20)), int(rng.integers(20, w - 20))] = 255
        return img
    if kind == "noise":            # white noise: candidates everywhere
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind in ("tie_cluster", "tie_clusters"):
        # black frame with one or two tight clusters of identical blobs (3 x 3 of 180 around a 255 centre: one corner of
        # score 254 each).  All keys of a cluster fall into one quadrant round after round, so the node list stops
        # growing at once (orb_feature_extractor.cpp:422) and the final node holds equal responses; the cluster
        # straddles a depth-2 quadrant boundary (x = 152 + 16) inside one FAST cell, with its first corner in candidate
        # order to the right of it -- a key order by deeper quadrant code would put another corner first
        img = np.zeros((h, w), np.uint8)
        pts = [(170, 91), (179, 91), (153, 100), (162, 100), (171, 100), (150, 109), (159, 109), (168, 109)]
        if kind == "tie_clusters":
            pts = pts + [(x + w // 2 - 20, y + h // 2 + 10) for x, y in pts]
        for x, y in pts:
            img[y - 1:y + 2, x - 1:x + 2] = 180
            img[y, x] = 255
        return img
    return synth.frame(seed, h, w)


@pytest.mark.parametrize("w,h,nfeat,scale,levels,fi,fm,kind,seed", [
    (640, 480, 1000, 1.5, 5, 20, 7, "rich", 41),        # coarse pyramid
    (640, 480, 1000, 1.1, 12, 20, 7, "rich", 42),       # fine pyramid, 12 levels
    (640, 480, 500, 1.2, 1, 20, 7, "rich", 43),         # a single level
    (333, 251, 600, 1.2, 6, 20, 7, "rich", 44),         # odd sizes: unaligned rows, partial tiles everywhere
    (1001, 517, 2000, 1.3, 7, 30, 10, "rich", 45),      # other FAST thresholds
    (640, 480, 1000, 1.2, 8, 12, 5, "rich", 46),
    (640, 480, 1000, 1.2, 8, 20, 7, "checker", 47),
    (640, 480, 1000, 1.2, 8, 20, 7, "saturated", 48),
    (640, 480, 1000, 1.2, 8, 20, 7, "gradient", 49),
    (3840, 2160, 5000, 1.2, 8, 20, 7, "rich", 50),      # BASELINE.json configs[2], 4K
    (640, 480, 3000, 1.2, 1, 12, 5, "rich", 51),        # 3000 keypoints on ONE level: quadtree node lists in global scratch
    (1280, 960, 12000, 1.2, 2, 12, 5, "noise", 52),     # the same with 1024 threads and two levels
    (640, 480, 1960, 1.2, 1, 12, 5, "noise", 53),       # node lists just below the LDS limit (about 154 KB of dynamic LDS)
    (640, 480, 2040, 1.2, 1, 12, 5, "noise", 54),       # ... and just above it
    (900, 700, 800, 1.9, 3, 20, 7, "rich", 55),         # steep pyramids: 1.9 still takes the strip resize kernel,
    (900, 700, 800, 2.2, 3, 20, 7, "rich", 56),         # 2.2 and 2.35 the LDS-tiled one (a pixel quad reaches more
    (1000, 800, 800, 2.35, 3, 20, 7, "noise", 57),      # than 8 source bytes)
])
def test_extract_parameter_sweep(oracle, w, h, nfeat, scale, levels, fi, fm, kind, seed):
    """other constructor arguments, image sizes and degenerate scenes: keypoints and descriptors against the oracle"""
    from orb_slam3_amd.features import ORBFeatureExtractor, Features
    img = _scene(kind, seed, h, w)
    p = oracle.params(w, h, scale, levels, fi, fm)
    okp, odesc = oracle.extract_literal(p, img, nfeat)
    fe = ORBFeatureExtractor(w, h, scale, levels, fi, fm, max_features=nfeat)
    ft = Features(w, h)
    assert fe.Extract(img, ft, nfeat) == 0
    _kp_equal(ft.keypoints, okp)
    assert np.array_equal(ft.descriptors, odesc)
    fe.close()


def _fuzz_case(seed):
    """a random but reproducible constructor configuration whose top level still holds a 30-px FAST cell"""
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(150, 900)), int(rng.integers(150, 700))
    if h > 2 * w:
        h = 2 * w
    scale = float(rng.choice([1.15, 1.2, 1.25, 1.33, 1.5, 1.7]))
    levels = 1
    while levels < 10 and min(w, h) / scale ** levels >= 70:
        levels += 1
    levels = int(rng.integers(1, levels + 1))
    fm = int(rng.integers(3, 15))
    fi = int(rng.integers(fm, 40))
    nfeat = int(rng.integers(20, 2500))
    kind = str(rng.choice(["rich", "rich", "low", "noise"]))
    return w, h, nfeat, scale, levels, fi, fm, kind


@pytest.mark.parametrize("seed", list(range(100, 116)))
def test_extract_random_configurations(oracle, seed):
    """sixteen seeded random configurations (size, pyramid, thresholds, budget, scene) against the oracle"""
    from orb_slam3_amd.features import ORBFeatureExtractor, Features
    w, h, nfeat, scale, levels, fi, fm, kind = _fuzz_case(seed)
    img = synth.frame(seed, h, w)
    if kind == "low":
        img = (img.astype(np.float32) * 0.12 + 100).astype(np.uint8)
    elif kind == "noise":
        img = np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)
    cfg = (w, h, nfeat, scale, levels, fi, fm, kind)
    p = oracle.params(w, h, scale, levels, fi, fm)
    try:
        okp, odesc = oracle.extract(p, img, nfeat)
    except ValueError:
        # a geometry the reference cannot run either (no quadtree root when the height is about twice the width):
        # the library must refuse it at construction
        from orb_slam3_amd._lib import OrbGpuError
        with pytest.raises(OrbGpuError):
            ORBFeatureExtractor(w, h, scale, levels, fi, fm, max_features=nfeat)
        return
    fe = ORBFeatureExtractor(w, h, scale, levels, fi, fm, max_features=nfeat)
    ft = Features(w, h)
    assert fe.Extract(img, ft, nfeat) == 0
    assert ft.Size() == len(okp), cfg
    for fld in ("x", "y", "level", "size"):
        assert np.array_equal(ft.keypoints[fld], okp[fld]), (fld, cfg)
    assert np.array_equal(ft.keypoints["angle"], okp["angle"].astype(np.float32)), cfg
    assert np.array_equal(ft.descriptors, odesc), cfg
    fe.close()


@pytest.mark.parametrize("fi,fm", [(1, 1), (2, 1), (3, 2), (4, 3), (20, 20), (23, 19), (100, 60), (254, 200), (255, 255)])
@pytest.mark.parametrize("kind", ["rich", "noise", "steps"])
def test_fast_threshold_extremes(oracle, fi, fm, kind):
    """the quick test of k_fast_cells works on pixels quantised to 6 bits with thresholds (t + 1) >> 2: thresholds at
    both ends of the range and around multiples of 4, on a scene, on noise and on saturated steps (0 / 255 edges, where
    the byte-wise sums sit at the ends of their ranges)"""
    from orb_slam3_amd.features import ORBFeatureExtractor, Features
    w, h, nfeat = 333, 261, 600
    if kind == "rich":
        img = synth.frame(4242, h, w)
    elif kind == "noise":
        img = np.random.default_rng(fi * 256 + fm).integers(0, 256, (h, w), dtype=np.uint8)
    else:
        rng = np.random.default_rng(99)
        img = np.zeros((h, w), np.uint8)
        for _ in range(400):
            y, x = int(rng.integers(0, h - 6)), int(rng.integers(0, w - 6))
            img[y:y + int(rng.integers(1, 12)), x:x + int(rng.integers(1, 12))] = int(rng.choice([0, 1, 3, 4, 127, 128, 251, 252, 254, 255]))
    p = oracle.params(w, h, 1.2, 4, fi, fm)
    okp, odesc = oracle.extract(p, img, nfeat)
    fe = ORBFeatureExtractor(w, h, 1.2, 4, fi, fm, max_features=nfeat)
    ft = Features(w, h)
    assert fe.Extract(img, ft, nfeat) == 0
    assert ft.Size() == len(okp)
    for fld in ("x", "y", "level", "size"):
        assert np.array_equal(ft.keypoints[fld], okp[fld]), fld
    assert np.array_equal(ft.keypoints["angle"], okp["angle"].astype(np.float32))
    assert np.array_equal(ft.descriptors, odesc)
    fe.close()


def test_resize_kernels_agree(oracle, monkeypatch):
    """the LDS-tiled resize kernel (the fallback for steep scale factors) forced for an ordinary pyramid: the same
    keypoints and descriptors as the strip kernel and the oracle"""
    from orb_slam3_amd.features import ORBFeatureExtractor, Features
    w, h, nfeat = 517, 389, 700
    img = synth.frame(77, h, w)
    okp, odesc = oracle.extract_literal(oracle.params(w, h, 1.2, 8, 20, 7), img, nfeat)
    for mode in ("tiles", None):
        if mode:
            monkeypatch.setenv("ORBGPU_TUNE", "resize=" + mode)
        else:
            monkeypatch.delenv("ORBGPU_TUNE", raising=False)
        fe = ORBFeatureExtractor(w, h, 1.2, 8, 20, 7, max_features=nfeat)
        ft = Features(w, h)
        assert fe.Extract(img, ft, nfeat) == 0
        _kp_equal(ft.keypoints, okp)
        assert np.array_equal(ft.descriptors, odesc)
        fe.close()


@pytest.mark.parametrize("w,h,nfeat,levels,seed,scene", [
    (640, 480, 1000, 8, 31, "rich"),       # nIni = 1, three presorted rounds on level 0
    (640, 480, 30, 8, 32, "rich"),         # tiny budgets: one presorted round, the list stops right behind it
    (640, 480, 6000, 8, 33, "rich"),       # budget above the candidates: every node ends as a singleton
    (1280, 720, 3000, 8, 34, "rich"),      # nIni = 2, keys in the global buffers
    (800, 300, 1200, 5, 35, "rich"),       # nIni = 3: a code range that is not a power of two
    (1100, 200, 900, 3, 36, "rich"),       # nIni = 6: those levels are not presorted, the others are
    (640, 480, 1000, 8, 37, "checker"),    # exact ties: equal responses inside a node (first maximum by candidate order)
    (640, 480, 2000, 8, 38, "noise"),      # cells with far more than 16 candidates
    (640, 480, 800, 8, 39, "gradient"),    # most cells and most codes empty
    (640, 480, 1000, 8, 40, "tie_cluster"),   # phase 1 stops before the presorted depth: ties inside a node whose keys are
    (640, 480, 1000, 8, 41, "tie_clusters"),  # ordered deeper than the node (first maximum must follow candidate order)
    (1920, 1080, 5000, 8, 42, "tie_clusters"),
])
def test_extract_quadtree_presorted(oracle, monkeypatch, w, h, nfeat, levels, seed, scene):
    """the presort of the quadtree's first rounds (k_qt_presort, by default for levels of 0.4 Mpx and more) forced for
    every level: keypoints and descriptors against the oracle, and identical to the plain kernel"""
    from orb_slam3_amd.features import ORBFeatureExtractor, Features
    img = _scene(scene, seed, h, w)
    okp, odesc = oracle.extract_literal(oracle.params(w, h, levels=levels), img, nfeat)
    for minpx in ("0", "-1"):
        monkeypatch.setenv("ORBGPU_TUNE", "qt_presort_minpx=" + minpx)
        fe = ORBFeatureExtractor(w, h, 1.2, levels, 20, 7, max_features=nfeat)
        ft = Features(w, h)
        assert fe.Extract(img, ft, nfeat) == 0
        _kp_equal(ft.keypoints, okp)
        assert np.array_equal(ft.descriptors, odesc)
        fe.close()


@pytest.mark.parametrize("w,h,plan", [(640, 480, "4:3"), (640, 480, "3:2,5:2"), (517, 389, "2:3,5:2"),
                                      (752, 480, "1:3,4:3"), (320, 240, "0:2")])
def test_resize_nest_agrees(oracle, monkeypatch, w, h, plan):
    """two or three pyramid levels per launch (k_resize_nest, opt-in): the pyramid levels are byte-identical to the
    oracle's, and so are keypoints and descriptors"""
    from orb_slam3_amd.features import ORBFeatureExtractor, Features
    levels = 8 if w >= 500 else 5
    img = synth.frame(91, h, w)
    p = oracle.params(w, h, 1.2, levels, 20, 7)
    okp, odesc, ex = oracle.extract(p, img, 800, dump=True)
    monkeypatch.setenv("ORBGPU_TUNE", "resize_nest=" + plan)
    fe = ORBFeatureExtractor(w, h, 1.2, levels, 20, 7, max_features=800)
    ft = Features(w, h)
    assert fe.Extract(img, ft, 800) == 0
    g = ex["geometry"]
    for l, (a, b) in enumerate(zip(_levels(fe.debug_dump()["pyramid"], g), _levels(ex["pyramid"], g))):
        assert np.array_equal(a, b), f"pyramid level {l}"
    _kp_equal(ft.keypoints, okp)
    assert np.array_equal(ft.descriptors, odesc)
    fe.close()


def test_single_frame_graph_recapture_and_eager(oracle, monkeypatch):
    """orbgpu_extract replays ONE captured hipGraph per feature_count (results land in mapped host memory): a context asked
    for 500, then 1000, then 500 keypoints again re-captures each time and stays bit-exact, the frame changes between
    replays of one graph, and ORBGPU_TUNE=no_graph=1 (eager launches, copies back) gives the same."""
    from orb_slam3_amd.features import ORBFeatureExtractor, Features
    w, h = 640, 480
    frames = synth.sequence(61, 3, h, w)
    p = oracle.params(w, h)
    want = {(f, nf): oracle.extract(p, frames[f], nf) for f in range(3) for nf in (500, 1000)}
    for no_graph in (None, "1"):
        if no_graph:
            monkeypatch.setenv("ORBGPU_TUNE", "no_graph=" + no_graph)
        fe = ORBFeatureExtractor(w, h, 1.2, 8, 20, 7, max_features=1000)
        for f, nf in ((0, 500), (1, 500), (2, 1000), (0, 1000), (1, 500), (2, 500), (0, 1000)):
            ft = Features(w, h)
            assert fe.Extract(frames[f], ft, nf) == 0
            okp, odesc = want[(f, nf)]
            _kp_equal(ft.keypoints, okp)
            same = np.array_equal(ft.descriptors, odesc)
            assert same, (no_graph, f, nf)
        fe.close()


@pytest.mark.parametrize("w,h,nfeat,levels,seed,scene", [
    (640, 480, 1000, 8, 71, "rich"),
    (333, 251, 600, 6, 72, "rich"),        # odd sizes: unaligned rows; small levels where most keypoints sit near a border
    (160, 120, 400, 3, 73, "noise"),       # nearly every patch reaches beyond the image: the reflected byte-wise gather
    (1280, 720, 3000, 8, 74, "rich"),
])
def test_patch_blur_mode(oracle, monkeypatch, w, h, nfeat, levels, seed, scene):
    """ORBGPU_TUNE=patch_blur=1 forces the per-keypoint path (k_patch_describe: orientation + 7 x 7 blur of the keypoint's own
    43 x 43 patch + rBRIEF in one kernel, by default only where the patches cover well under the pyramid, e.g. 4K / 5000):
    keypoints, angles and descriptors against the oracle -- i.e. against the full-frame cv::GaussianBlur with
    BORDER_REFLECT_101 -- and the blurred levels of the stage dump (produced on demand in this mode) as well."""
    from orb_slam3_amd.features import ORBFeatureExtractor, Features
    img = _scene(scene, seed, h, w)
    p = oracle.params(w, h, levels=levels)
    okp, odesc, ex = oracle.extract(p, img, nfeat, dump=True)
    monkeypatch.setenv("ORBGPU_TUNE", "patch_blur=1")
    for max_batch in (1, 2):               # the captured single-frame graph and the eager batch path
        fe = ORBFeatureExtractor(w, h, 1.2, levels, 20, 7, max_batch=max_batch, max_features=nfeat)
        ft = Features(w, h)
        assert fe.Extract(img, ft, nfeat) == 0
        _kp_equal(ft.keypoints, okp)
        same = np.array_equal(ft.descriptors, odesc)
        assert same, max_batch
        d = fe.debug_dump()
        g = ex["geometry"]
        for l, (a, b) in enumerate(zip(_levels(d["blurred"], g), _levels(ex["blurred"], g))):
            if ex["n_kp"][l]:
                assert np.array_equal(a, b), f"blur level {l}"
        fe.close()
    near = ((okp["x"] / 1.2 ** okp["level"] < 21.5) | (okp["y"] / 1.2 ** okp["level"] < 21.5)).sum()
    assert near > 0                        # some patches did reach beyond the image
