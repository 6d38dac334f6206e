"""CPU tests that pin the oracle (oracle/) against everything the reference offers for this path:
the derived geometry of SURVEY.md App. C, the reference's own known-answer test for the window
query (test/features/feature_tests.cpp:21-50) and the self-consistency checks of SURVEY.md 8(c)."""
import math

import numpy as np
import pytest

from orb_slam3_amd import synth

APP_C_640 = [  # (w, h, budget, ncols, nrows, wcell, hcell, nini, kp_size) -- SURVEY.md App. C
    (640, 480, 217, 20, 14, 31, 32, 1, 31), (533, 400, 181, 16, 12, 32, 31, 1, 37),
    (444, 333, 151, 13, 10, 32, 31, 1, 44), (370, 278, 126, 11, 8, 31, 31, 1, 53),
    (309, 231, 105, 9, 6, 31, 34, 1, 64), (257, 193, 87, 7, 5, 33, 33, 1, 77),
    (214, 161, 73, 6, 4, 31, 33, 1, 92), (179, 134, 60, 4, 3, 37, 34, 1, 111)]


def test_geometry_640(oracle):
    g = oracle.geometry(oracle.params(640, 480), 1000)
    got = [(g.w[l], g.h[l], g.budget[l], g.ncols[l], g.nrows[l], g.wcell[l], g.hcell[l], g.nini[l], g.kp_size[l])
           for l in range(8)]
    assert got == APP_C_640
    assert sum(g.w[l] * g.h[l] for l in range(8)) == 950532


def test_geometry_other_configs(oracle):
    g = oracle.geometry(oracle.params(512, 512), 1500)
    assert [g.w[l] for l in range(8)] == [512, 427, 356, 296, 247, 206, 171, 143]
    assert [g.budget[l] for l in range(8)] == [326, 271, 226, 189, 157, 131, 109, 91]
    g = oracle.geometry(oracle.params(512, 512), 7500)
    assert [g.budget[l] for l in range(8)] == [1629, 1357, 1131, 943, 785, 655, 545, 455]
    g = oracle.geometry(oracle.params(1920, 1080), 5000)
    assert [g.budget[l] for l in range(8)] == [1086, 905, 754, 628, 524, 436, 364, 303]
    assert [(g.w[l], g.h[l]) for l in range(8)] == [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521),
                                                    (772, 434), (643, 362), (536, 301)]
    assert [(g.ncols[l], g.nrows[l], g.wcell[l], g.hcell[l]) for l in (0, 7)] == [(62, 34, 31, 31), (16, 8, 32, 34)]
    assert g.nini[0] == 2
    assert sum(g.w[l] * g.h[l] for l in range(8)) == 6419321
    g = oracle.geometry(oracle.params(3840, 2160), 5000)
    assert [(g.w[l], g.h[l]) for l in range(8)] == [(3840, 2160), (3200, 1800), (2667, 1500), (2222, 1250),
                                                    (1852, 1042), (1543, 868), (1286, 723), (1072, 603)]
    assert (g.ncols[0], g.nrows[0]) == (126, 70)
    assert sum(g.w[l] * g.h[l] for l in range(8)) == 25677702


def test_list_features_in_area_reference_known_answer(oracle):
    # keypoints and queries of /root/reference/test/features/feature_tests.cpp:21-50
    pts = [(120, 62, 0), (20, 42, 1), (57, 259, 0), (185, 336, 2), (198, 78, 0), (1, 1, 0), (17, 451, 5),
           (301, 42, 0), (510, 79, 1), (604, 21, 0), (632, 262, 0)]
    kps = np.zeros(len(pts), oracle.KP_DTYPE)
    for i, (x, y, l) in enumerate(pts):
        kps[i]["x"], kps[i]["y"], kps[i]["level"] = x, y, l
    assert len(oracle.list_features_in_area(kps, 190, 80, 10, 0, 2)) == 1
    assert len(oracle.list_features_in_area(kps, 190, 80, 10, 1, 2)) == 0
    r = oracle.list_features_in_area(kps, 190, 80, 100, 0, 2)
    assert len(r) == 2 and list(r) == [0, 4]
    # level rule of features.cpp:94,109-114: (-1,-1) disables the level check, max<0 means open-ended
    assert len(oracle.list_features_in_area(kps, 0, 0, 10000, -1, -1)) == len(pts)
    assert list(oracle.list_features_in_area(kps, 0, 0, 10000, 2, -1)) == [3, 6]
    # strict inequality on the window edge (features.cpp:119)
    assert len(oracle.list_features_in_area(kps, 188, 78, 10, 0, 0)) == 0
    assert list(oracle.list_features_in_area(kps, 189, 78, 10, 0, 0)) == [4]


@pytest.mark.parametrize("seed,shape", [(1, (40, 52)), (2, (64, 64)), (3, (37, 91))])
def test_fast_literal_equals_closed_form(oracle, seed, shape):
    rng = np.random.default_rng(seed)
    # mix of flat-ish regions and strong edges so both thresholds fire
    img = (rng.integers(0, 4, shape) * 60 + rng.integers(0, 30, shape)).astype(np.uint8)
    for thr in (7, 20, 40):
        x, y, s = oracle.fast_literal(img, thr, True)
        smap = oracle.fast_score_map(img, thr).astype(np.int32)
        h, w = shape
        exp = []
        for yy in range(3, h - 3):
            for xx in range(3, w - 3):
                v = smap[yy, xx]
                if v < thr:
                    continue
                nb = smap[yy - 1:yy + 2, xx - 1:xx + 2].copy()
                nb[1, 1] = -1
                if (v > nb).all():
                    exp.append((xx, yy, v))
        assert list(zip(x.tolist(), y.tolist(), s.tolist())) == exp
        assert len(exp) > 0
        # without NMS every pixel whose closed-form score >= thr is a corner, and only those
        x2, y2, _ = oracle.fast_literal(img, thr, False)
        mask = np.zeros(shape, bool)
        mask[y2, x2] = True
        assert np.array_equal(mask, smap >= thr)


def test_detect_level_literal_equals_scoremap(oracle):
    img = synth.frame(11, 200, 260)
    a = oracle.detect_level(img, 20, 7, literal=False)
    b = oracle.detect_level(img, 20, 7, literal=True)
    assert len(a) > 50 and np.array_equal(a, b)
    # low-contrast image: exercises the per-cell fallback to the min threshold
    low = (img.astype(np.int32) // 6 + 100).astype(np.uint8)
    a = oracle.detect_level(low, 20, 7, literal=False)
    b = oracle.detect_level(low, 20, 7, literal=True)
    assert len(a) > 10 and np.array_equal(a, b)
    assert a["response"].min() < 20


def test_gaussian_blur(oracle):
    const = np.full((40, 50), 137, np.uint8)
    assert (oracle.gaussian_blur7(const) == 137).all()
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (48, 70)).astype(np.uint8)
    out = oracle.gaussian_blur7(img)
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.float64) / 256
    assert abs(k.sum() - 1) < 1e-12
    pad = np.pad(img.astype(np.float64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    hh = sum(k[i] * pad[:, i:i + 70] for i in range(7))
    vv = sum(k[j] * hh[j:j + 48, :] for j in range(7))
    assert np.array_equal(out, np.floor(vv + 0.5).astype(np.uint8))
    # the taps are within rounding of the true sigma=2 Gaussian
    g = np.exp(-np.arange(-3, 4) ** 2 / 8.0)
    assert np.abs(k * 256 - g / g.sum() * 256).max() < 1.0


def test_resize_close_to_exact_bilinear(oracle):
    rng = np.random.default_rng(6)
    src = rng.integers(0, 256, (100, 120)).astype(np.uint8)
    dw, dh = 100, 83
    out = oracle.resize_linear(src, dw, dh).astype(np.float64)
    sx = (np.arange(dw) + 0.5) * (120 / dw) - 0.5
    sy = (np.arange(dh) + 0.5) * (100 / dh) - 0.5
    x0 = np.clip(np.floor(sx).astype(int), 0, 118)
    y0 = np.clip(np.floor(sy).astype(int), 0, 98)
    fx = np.clip(sx - x0, 0, 1)
    fy = np.clip(sy - y0, 0, 1)[:, None]
    s = src.astype(np.float64)
    ref = (s[y0][:, x0] * (1 - fx) + s[y0][:, x0 + 1] * fx) * (1 - fy) + (s[y0 + 1][:, x0] * (1 - fx) + s[y0 + 1][:, x0 + 1] * fx) * fy
    assert np.abs(out - ref).max() <= 1.0
    # identity resize is exact
    assert np.array_equal(oracle.resize_linear(src, 120, 100), src)
    # upscale exercises the sx<0 / xmax tail branches; constant image must stay constant
    c = np.full((20, 30), 77, np.uint8)
    assert (oracle.resize_linear(c, 47, 33) == 77).all()


def test_fast_atan2(oracle):
    rng = np.random.default_rng(7)
    for _ in range(2000):
        y, x = rng.normal(size=2) * 1000
        ref = math.degrees(math.atan2(y, x)) % 360.0
        got = oracle.fast_atan2(y, x)
        d = abs(got - ref)
        assert min(d, 360 - d) < 0.3
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
    assert abs(oracle.fast_atan2(1.0, 0.0) - 90.0) < 1e-3


def test_hamming(oracle):
    rng = np.random.default_rng(8)
    for _ in range(200):
        a = rng.integers(0, 256, 32).astype(np.uint8)
        b = rng.integers(0, 256, 32).astype(np.uint8)
        assert oracle.hamming256(a, b) == sum(bin(int(v)).count("1") for v in (a ^ b))
    z = np.zeros(32, np.uint8)
    assert oracle.hamming256(z, ~z) == 256 and oracle.hamming256(z, z) == 0


def _snn_py(to_desc, from_desc, cands, th, ratio):
    """Literal python transcription of MatchWithIterators' control flow with dicts
    (second_nearest_neighbor_matcher.hpp:30-80)."""
    UMAX = 2 ** 32 - 1
    out, from_to, best_from = {}, {}, {}
    for i, cj in cands:
        best = second = UMAX
        bj = None
        for j in cj:
            d = sum(bin(int(v)).count("1") for v in (to_desc[i] ^ from_desc[j]))
            if d > th:
                continue
            if d < best:
                bj, second, best = j, best, d
            elif d < second:
                second = d
        if best < ratio * second:
            if bj in best_from:
                if best_from[bj] > best:
                    del out[from_to[bj]]
                else:
                    continue
            best_from[bj] = best
            from_to[bj] = i
            out[i] = bj
    return out


def test_snn_bruteforce_matches_literal(oracle):
    rng = np.random.default_rng(9)
    frm = rng.integers(0, 256, (60, 32)).astype(np.uint8)
    to = synth.descriptors(10, 80, planted_from=frm, planted_frac=0.7, max_flips=60)
    to[5] = to[4]  # two 'to' rows claim the same 'from' with equal distance: earliest keeps it
    frm[11] = frm[10]  # tie between best and second: ratio test must fail
    for th, ratio in ((50, 0.9), (100, 0.8), (50, 1.0), (100, 0.6), (256, 0.9)):
        mf, md = oracle.snn_match(to, frm, th, ratio)
        exp = _snn_py(to, frm, [(i, range(60)) for i in range(80)], th, ratio)
        got = {i: int(j) for i, j in enumerate(mf) if j >= 0}
        assert got == exp
        assert len(set(got.values())) == len(got)
    assert sum(1 for v in got.values()) > 5


def test_snn_window_and_area(oracle):
    rng = np.random.default_rng(12)
    n = 70
    kf = np.zeros(n, oracle.KP_DTYPE)
    kt = np.zeros(n, oracle.KP_DTYPE)
    kf["x"] = rng.integers(0, 300, n); kf["y"] = rng.integers(0, 200, n); kf["level"] = rng.integers(0, 3, n)
    kt["x"] = kf["x"] + rng.integers(-5, 6, n); kt["y"] = kf["y"] + rng.integers(-5, 6, n); kt["level"] = kf["level"]
    kt["level"][0] = 2
    df = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    dt = df.copy()
    dt[:, 0] ^= rng.integers(0, 256, n).astype(np.uint8)
    mf, md = oracle.match_area(kt, dt, kf, df, window=30, threshold=50, ratio=0.9)
    cands = []
    for i in range(n):
        if i == 0 or kt["level"][i] == 0:
            cj = list(oracle.list_features_in_area(kf, kt["x"][i], kt["y"][i], 30, 0, 0))
            cands.append((i, cj))
    exp = _snn_py(dt, df, cands, 50, 0.9)
    assert {i: int(j) for i, j in enumerate(mf) if j >= 0} == exp and len(exp) > 5


def _orient_py(ang_to, ang_from, matches):
    hist = [[] for _ in range(30)]
    factor = float(np.float32(1.0) / np.float32(30))
    for i, j in matches.items():
        rot = np.float32(ang_to[i] - ang_from[j])
        if rot < 0.0:
            rot = np.float32(rot + np.float32(360.0))
        v = float(rot) * factor
        b = int(math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1))
        if b == 30:
            b = 0
        hist[b].append(i)
    max1 = max2 = max3 = 0
    i1 = i2 = i3 = -1
    for i in range(30):
        s = len(hist[i])
        if s > max1:
            max3, max2, max1, i3, i2, i1 = max2, max1, s, i2, i1, i
        elif s > max2:
            max3, max2, i3, i2 = max2, s, i2, i
        elif s > max3:
            max3, i3 = s, i
    if np.float32(max2) < np.float32(0.1) * np.float32(max1):
        i2 = i3 = -1
    elif np.float32(max3) < np.float32(0.1) * np.float32(max1):
        i3 = -1
    keep = dict(matches)
    nd = 0
    for b in range(30):
        if b in (i1, i2, i3):
            continue
        for i in hist[b]:
            del keep[i]
            nd += 1
    return keep, nd


def test_orientation_validator(oracle):
    rng = np.random.default_rng(13)
    n = 300
    a_from = np.float32(rng.uniform(0, 360, n)).astype(np.float64)
    a_to = np.float32((a_from + rng.normal(20, 12, n)) % 360).astype(np.float64)
    a_to[:40] = np.float32(rng.uniform(0, 360, 40))
    mf = np.arange(n, dtype=np.int32)
    mf[rng.random(n) < 0.2] = -1
    matches = {i: int(j) for i, j in enumerate(mf) if j >= 0}
    exp, nd = _orient_py(a_to, a_from, matches)
    got, gnd = oracle.orientation_validate(a_to, a_from, mf)
    assert gnd == nd and {i: int(j) for i, j in enumerate(got) if j >= 0} == exp
    assert 0 < nd < len(matches)


def test_extract_invariants(oracle):
    p = oracle.params(320, 240)
    img = synth.frame(21, 240, 320)
    kps, desc, ex = oracle.extract(p, img, 500, dump=True)
    g = ex["geometry"]
    assert len(kps) == sum(ex["n_kp"]) and len(kps) >= 400
    # reversed packing (App. B.4): levels descend with index
    assert (np.diff(kps["level"]) <= 0).all()
    assert kps["level"][0] == max(l for l in range(8) if ex["n_kp"][l]) and kps["level"][-1] == 0
    for l in range(8):
        m = kps["level"] == l
        assert m.sum() == ex["n_kp"][l] <= g.budget[l] + 3
        s = float(np.float32(g.scale[l]))
        lx, ly = kps["x"][m] / (s if l else 1.0), kps["y"][m] / (s if l else 1.0)
        assert (np.abs(lx - np.round(lx)) < 1e-9).all()
        assert lx.min() >= 19 - 1e-9 and lx.max() <= g.w[l] - 20 + 1e-9
        assert ly.min() >= 19 - 1e-9 and ly.max() <= g.h[l] - 20 + 1e-9
        assert (kps["size"][m] == g.kp_size[l]).all()
    assert ((kps["angle"] >= 0) & (kps["angle"] < 360)).all()
    k2, d2 = oracle.extract_literal(p, img, 500)
    assert np.array_equal(kps, k2) and np.array_equal(desc, d2)
    # descriptor is recomputable from the dumped blurred level
    i = len(kps) - 1
    blur0 = ex["blurred"][:320 * 240].reshape(240, 320)
    assert np.array_equal(desc[i], oracle.orb_descriptor(blur0, int(kps["x"][i]), int(kps["y"][i]), kps["angle"][i]))


def test_extract_edge_cases(oracle):
    p = oracle.params(160, 120, levels=3)
    flat = np.full((120, 160), 90, np.uint8)
    kps, desc = oracle.extract(p, flat, 100)
    assert len(kps) == 0 and desc.shape == (0, 32)
    with pytest.raises(ValueError):
        oracle.geometry(oracle.params(64, 64, levels=8), 100)  # top levels smaller than one 30-px cell


def test_predict_scale(oracle):
    ls = math.log(1.2)
    for d, m in ((1.0, 1.0), (1.0, 1.3), (2.0, 1.0), (1.0, 100.0), (3.0, 4.0)):
        e = max(0.0, math.ceil(math.log(m / d) / ls))
        e = min(int(e), 7)
        assert oracle.predict_scale(d, m, 1.2, 8) == e


def _literal_bow_match(desc_to, desc_from, fv_to, fv_from, threshold, ratio, ham):
    """Line-by-line transcription of BowToIterator (bow_to_iterator.cpp:12-47) driving
    SNNMatcher::MatchWithIterators (second_nearest_neighbor_matcher.hpp:30-80) on dict-based feature vectors."""
    import bisect
    to_nodes, from_nodes = sorted(fv_to), sorted(fv_from)
    kt = kf = 0
    matches, best_matches, best_dist = {}, {}, {}
    while kt < len(to_nodes) and kf < len(from_nodes):      # AdvanceUntilSameNode
        if to_nodes[kt] == from_nodes[kf]:
            ids = fv_to[to_nodes[kt]]
            if not ids:
                kt += 1
                kf += 1
                continue
            for i in ids:                                     # operator++ inside one node
                best = second = 0xFFFFFFFF
                best_j = None
                for j in fv_from[from_nodes[kf]]:
                    d = ham(desc_to[i], desc_from[j])
                    if d > threshold:
                        continue
                    if d < best:
                        best_j, second, best = j, best, d
                    elif d < second:
                        second = d
                if best < ratio * second:
                    if best_j in best_matches:
                        if best_dist[best_j] > best:
                            del matches[best_matches[best_j]]
                        else:
                            continue
                    best_dist[best_j] = best
                    best_matches[best_j] = i
                    matches[i] = best_j
            kt += 1
            kf += 1
        elif to_nodes[kt] < from_nodes[kf]:
            kt = bisect.bisect_left(to_nodes, from_nodes[kf])
        else:
            kf = bisect.bisect_left(from_nodes, to_nodes[kt])
    return matches


def test_match_bow_vs_literal(oracle):
    from orb_slam3_amd import synth
    rng = np.random.default_rng(21)
    for trial in range(6):
        n_from, n_to = int(rng.integers(40, 400)), int(rng.integers(40, 400))
        frm = synth.descriptors(100 + trial, n_from)
        to = synth.descriptors(200 + trial, n_to, planted_from=frm, planted_frac=0.6)
        n_nodes = int(rng.integers(3, 40))
        # node of a feature: mostly a function of the descriptor (so planted pairs share a node), some noise, some
        # nodes present on one side only, and an occasional empty 'to' list
        node_of = lambda d: int(d[0]) % n_nodes
        fv_to, fv_from = {}, {}
        for i in range(n_to):
            nd = node_of(to[i]) if rng.random() < 0.9 else int(rng.integers(0, n_nodes))
            fv_to.setdefault(2 * nd + (1 if rng.random() < 0.05 else 0), []).append(i)
        for j in range(n_from):
            nd = node_of(frm[j]) if rng.random() < 0.9 else int(rng.integers(0, n_nodes))
            fv_from.setdefault(2 * nd + (1 if rng.random() < 0.05 else 0), []).append(j)
        if trial % 2:
            fv_to[2 * n_nodes + 4] = []
            fv_from[2 * n_nodes + 4] = [0, 1]
        # duplicate rows so that ties (best == second, equal-distance claims) occur
        to[1] = to[0]
        frm[3] = frm[2]
        ham = lambda a, b: int(np.unpackbits(np.bitwise_xor(a, b)).sum())
        for th, ratio in ((50, 0.6), (50, 0.7), (100, 0.9)):
            want = _literal_bow_match(to, frm, fv_to, fv_from, th, ratio, ham)
            mf, md = oracle.match_bow(to, frm, fv_to, fv_from, th, ratio)
            got = {i: int(j) for i, j in enumerate(mf) if j >= 0}
            assert got == want
            for i, j in got.items():
                assert md[i] == ham(to[i], frm[j])


def test_bow_transform_vs_literal(oracle):
    """oo_bow_transform against a literal transcription of TemplatedVocabulary::transform (single feature),
    TemplatedVocabulary.h:1218-1259, on ragged random trees with tied sibling descriptors."""
    rng = np.random.default_rng(8)
    ham = lambda a, b: int(np.unpackbits(np.bitwise_xor(a, b)).sum())
    for trial in range(4):
        k, L = int(rng.integers(2, 7)), int(rng.integers(1, 5))
        kids = [[]]
        frontier = [0]
        for lvl in range(1, L + 1):
            nxt = []
            for v in frontier:
                if lvl > 1 and rng.random() < 0.2:
                    continue
                for _ in range(int(rng.integers(1, k + 1))):
                    kids.append([])
                    kids[v].append(len(kids) - 1)
                    nxt.append(len(kids) - 1)
            frontier = nxt
        n = len(kids)
        for v in range(n):
            rng.shuffle(kids[v])
        nd = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        for v in range(n):
            if len(kids[v]) >= 2:
                nd[kids[v][-1]] = nd[kids[v][0]]
        co, ch = [0], []
        for v in range(n):
            ch.extend(kids[v])
            co.append(len(ch))
        wid = np.arange(n, dtype=np.uint32) * 3 + 1
        desc = rng.integers(0, 256, (200, 32), dtype=np.uint8)
        desc[:100] = nd[rng.integers(0, n, 100)]
        for levelsup in (0, 1, 3, 7):
            ow, on = oracle.bow_transform(desc, co, ch, nd, wid, L, levelsup)
            for i in range(len(desc)):
                nid_level = L - levelsup
                nid = 0
                final_id, current_level = 0, 0
                while True:
                    current_level += 1
                    nodes = kids[final_id]
                    final_id = nodes[0]
                    best_d = float(ham(desc[i], nd[final_id]))
                    for idn in nodes[1:]:
                        d = float(ham(desc[i], nd[idn]))
                        if d < best_d:
                            best_d, final_id = d, idn
                    if current_level == nid_level:
                        nid = final_id
                    if not kids[final_id]:
                        break
                assert ow[i] == wid[final_id] and on[i] == nid


def test_distinctive_descriptor_vs_literal(oracle):
    """oo_distinctive_descriptor against a transcription of map_point.cpp:135-171 (sort + element 0.5*(N-1))."""
    rng = np.random.default_rng(4)
    groups = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for n in (0, 1, 2, 3, 4, 7, 10, 33, 64, 65, 130)]
    groups[5][3] = groups[5][0]          # duplicates: equal medians, first row wins
    groups[6][:] = groups[6][0]          # all identical
    off = np.cumsum([0] + [len(g) for g in groups]).astype(np.int32)
    got = oracle.distinctive_descriptor(np.concatenate(groups), off)
    for p, g in enumerate(groups):
        N = len(g)
        if N == 0:
            assert got[p] == -1
            continue
        bits = np.unpackbits(g, axis=1).astype(np.int32)
        dist = (bits[:, None, :] != bits[None, :, :]).sum(-1)
        best_median, best_idx = 2 ** 31 - 1, 0
        for i in range(N):
            median = sorted(dist[i].tolist())[int(0.5 * (N - 1))]
            if median < best_median:
                best_median, best_idx = median, i
        assert got[p] == best_idx


def _py_undistort3(model, k, d):
    """camera/distortions/barrel5.cpp:44-70 and fish_eye.cpp:68-116 in Python floats (IEEE double, no FMA)"""
    import math

    def div(a, b):          # IEEE division (Python raises on a zero divisor)
        if b != 0:
            return a / b
        return math.nan if a == 0 or a != a else math.copysign(math.inf, a) * math.copysign(1.0, b)
    if model == 0:
        k1, k2, p1, p2, k3 = k[:5]
        x, y = d[0], d[1]
        x0, y0 = d[0], d[1]
        for _ in range(10):
            r2 = x * x + y * y
            icdist = div(1., 1 + ((k3 * r2 + k2) * r2 + k1) * r2)
            if icdist < 0:
                return False, (x, y, 1.0)
            dX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            xn, yn = (x0 - dX) * icdist, (y0 - dY) * icdist
            if abs(x - xn) < 1e-7 and abs(y - yn) < 1e-7:
                return True, (x, y, 1.0)
            x, y = xn, yn
        return False, (-10000.0, 10000.0, -1.0)
    k1, k2, k3, k4 = k[:4]
    theta_d = math.sqrt(d[0] * d[0] + d[1] * d[1])
    theta_d = min(math.pi / 2, max(-math.pi / 2, theta_d))
    theta, conv = theta_d, False
    for _ in range(10):
        g2 = theta * theta; g3 = g2 * theta; g4 = g3 * theta; g5 = g4 * theta
        g6 = g5 * theta; g7 = g6 * theta; g8 = g7 * theta; g9 = g8 * theta
        fp = 1 + 3 * k1 * g2 + 5 * k2 * g4 + 7 * k3 * g6 + 9 * k4 * g8
        fix = div(theta + k1 * g3 + k2 * g5 + k3 * g7 + k4 * g9 - theta_d, fp)
        theta -= fix
        conv = abs(fix) < 7e-8
        if conv:
            break
    if not conv:
        return False, (-10000.0, -10000.0, -1.0)
    scale = div(math.tan(theta), theta_d)
    flipped = (theta_d < 0 and theta > 0) or (theta_d > 0 and theta < 0)
    if not flipped and scale > 0:
        return True, (d[0] * scale, d[1] * scale, 1.0)
    return False, (-10000.0, -10000.0, -1.0)


CAMERAS = [
    (-1, ()),                                               # no distortion model
    (0, (-0.28, 0.07, 2e-4, 1.8e-5, 0.0)),                   # TUM-like radial-tangential
    (0, (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)),         # Freiburg1
    (0, (-80.0, 0.0, 0.0, 0.0, 0.0)),                         # absurd k1: icdist < 0 / no convergence on the rim
    (2, (0.0034, 0.0007, -0.0021, 0.0002)),                  # TUM-VI-like fisheye
    (2, (2.5, -9.0, 30.0, -40.0)),                            # ill-conditioned: some points do not converge
]


def test_keypoint_geometry_vs_literal(oracle):
    """oo_keypoint_geometry against a Python transcription of monocular_camera.cpp:31-61 + the distortion models."""
    rng = np.random.default_rng(2)
    n = 600
    kp = np.zeros(n, oracle.KP_DTYPE)
    kp["x"], kp["y"] = rng.uniform(0, 640, n) * 1.2 ** rng.integers(0, 3, n), rng.uniform(0, 480, n)
    kp["x"][:3], kp["y"][:3] = (320.5, 0, 639), (240.25, 0, 479)       # the principal point itself, corners
    for model, k in CAMERAS:
        fx, fy, cx, cy = 517.3, 516.5, 320.5, 240.25
        cam = oracle.camera(fx, fy, cx, cy, model, k)
        und, unp, ok = oracle.keypoint_geometry(kp, cam)
        fx_inv, fy_inv = 1 / fx, 1 / fy
        n_fail = 0
        for i in range(n):
            up = ((kp["x"][i] - cx) * fx_inv, (kp["y"][i] - cy) * fy_inv, 1.0)
            if model < 0:
                assert tuple(und[i]) == (kp["x"][i], kp["y"][i]) and tuple(unp[i]) == up and ok[i] == 3
                continue
            good, u = _py_undistort3(model, k, up)
            assert tuple(unp[i]) == u and ok[i] == (3 if good else 0)
            if good:
                z_inv = 1 / u[2]
                assert tuple(und[i]) == (u[0] * z_inv * fx + cx, u[1] * z_inv * fy + cy)
            else:
                n_fail += 1
                assert tuple(und[i]) == (0.0, 0.0)
        if k and abs(k[0]) > 2:
            assert n_fail > 0          # the failure branches are really exercised


@pytest.mark.parametrize("scoring", [0, 1, 2, 4, 5])
def test_bow_score_vs_literal(oracle, scoring):
    """oo_bow_score (inverted file + SearchWordSharingKeyFrames + vocabulary score) against the dict-based literal
    transcription of dbo_w2_database.cpp:53-132 and ScoringObject.cpp."""
    from db_literal import LiteralDatabase, flatten, l_score, random_database
    vocab = 500
    bows, freqs = random_database(40 + scoring, 60, vocab, 80, weighted_counts=True)
    db = LiteralDatabase(vocab)
    for k in bows:
        db.Append(k, bows[k], freqs[k])
    ids, offs, words, vals, counts = flatten(bows, freqs)
    for qk in (ids[0], ids[5], ids[17]):
        q_bow, q_wf = bows[qk], {w: c + 1 for w, c in freqs[qk].items()}
        qw = np.asarray(sorted(q_bow), np.uint32)
        qv = np.asarray([q_bow[w] for w in sorted(q_bow)], np.float64)
        qc = np.asarray([q_wf[w] for w in sorted(q_bow)], np.uint32)
        shared, score = oracle.bow_score(qw, qv, offs, words, vals, scoring, qc, counts)
        lit = db.SearchWordSharingKeyFrames(q_wf)
        for n, k in enumerate(ids):
            assert int(shared[n]) == lit.get(k, 0)
            s = l_score(q_bow, bows[k], scoring)
            assert score[n] == s or (math.isnan(score[n]) and math.isnan(s))
    # empty query, and counts left out (= 1)
    shared, score = oracle.bow_score(np.zeros(0, np.uint32), np.zeros(0), offs, words, vals, scoring)
    assert not shared.any()
    shared1, _ = oracle.bow_score(qw, qv, offs, words, vals, scoring)
    lit1 = db_counts_one(bows, q_bow)
    assert [int(x) for x in shared1] == [lit1[k] for k in ids]


def db_counts_one(bows, q_bow):
    return {k: len(set(bows[k]) & set(q_bow)) for k in bows}


def test_cosf_exposure_report(oracle):
    """orb_feature_extractor.cpp:124-125 calls cos / sin on a float: with <math.h> in scope that is cosf / sinf, which in
    glibc are within 1 ulp of, but not always equal to, the correctly rounded value the oracle and the kernel use.
    Over every keypoint angle of the bench batch's 32 frames: how many angles get a different cos or sin, and how many
    descriptor bits change with the other variant.  (Over ALL floats in [0, 2 pi], glibc 2.35: cosf differs for
    446,486 and sinf for 1,020,963 of 1,086,918,620; most of those floats are tiny, angles spread over the circle differ
    more often.)"""
    from orb_slam3_amd import synth
    frames = synth.sequence(1235, 32)
    p = oracle.params(640, 480)
    n_kp = n_angle_diff = n_desc_diff = n_bits = 0
    try:
        for f in frames:
            oracle.set_trig_variant(0)
            k0, d0 = oracle.extract(p, f, 1000)
            oracle.set_trig_variant(1)
            k1, d1 = oracle.extract(p, f, 1000)
            assert np.array_equal(k0, k1)                       # keypoints do not depend on it
            for a in k0["angle"]:
                n_angle_diff += oracle.descriptor_trig(a, 0) != oracle.descriptor_trig(a, 1)
            x = np.bitwise_xor(d0, d1)
            n_kp += len(k0)
            n_desc_diff += int((x.any(axis=1)).sum())
            n_bits += int(np.unpackbits(x).sum())
    finally:
        oracle.set_trig_variant(0)
    print(f"\ncosf exposure: {n_kp} keypoints, {n_angle_diff} angles with a different (cos, sin) pair, "
          f"{n_desc_diff} descriptors / {n_bits} bits differ")
    # measured here (glibc 2.35): 864 of 32,202 angles (2.7 %) get a cos or sin that differs by one ulp, and not one
    # descriptor bit changes: a 1-ulp change moves a tap coordinate by ~1e-6 px, far from every cvRound boundary
    assert n_angle_diff < 0.06 * n_kp
    assert n_bits <= 8


def test_working_grid_lists_the_same_set(oracle):
    """ListFeaturesInArea over the grid the reference means to have (PosInGrid's body behind the early return,
    features.cpp:143-161) returns the same SET as over the degenerate grid it ships with, including for windows that reach
    over the image border and keypoints outside the image; and the reference's known answer
    (test/features/feature_tests.cpp:21-50) holds over it too."""
    rng = np.random.default_rng(8)
    W, H = 752.0, 480.0
    n = 3000
    kps = np.zeros(n, oracle.KP_DTYPE)
    kps["x"] = rng.uniform(-30, W + 30, n)
    kps["y"] = rng.uniform(-30, H + 30, n)
    kps["level"] = rng.integers(0, 8, n)
    xy = np.stack([kps["x"], kps["y"]], 1)
    cs, items = oracle.assign_features_to_grid(xy, W, H)
    assert cs[0] == 0 and cs[-1] == n and (np.diff(cs) >= 0).all()
    assert sorted(items.tolist()) == list(range(n))
    for c in np.flatnonzero(np.diff(cs) > 1)[:200]:
        assert (np.diff(items[cs[c]:cs[c + 1]]) > 0).all()            # push_back order
    nonempty = 0
    for _ in range(400):
        px, py = rng.uniform(-50, W + 50), rng.uniform(-50, H + 50)
        win = int(rng.integers(0, 140))
        lmin, lmax = int(rng.integers(-1, 4)), int(rng.integers(-1, 8))
        flat = oracle.list_features_in_area(kps, px, py, win, lmin, lmax)
        grid = oracle.list_features_in_area_grid(xy, kps["level"], cs, items, W, H, px, py, win, lmin, lmax)
        assert sorted(grid.tolist()) == flat.tolist()
        nonempty += len(flat) > 0
    assert nonempty > 150
    # the reference's own fixture (test/features/feature_tests.cpp:21-50) over the working grid
    pts = [(120, 62, 0), (20, 42, 1), (57, 259, 0), (185, 336, 2), (198, 78, 0), (1, 1, 0), (17, 451, 5),
           (301, 42, 0), (510, 79, 1), (604, 21, 0), (632, 262, 0)]
    xy11 = np.array([(x, y) for x, y, _ in pts], np.float64)
    lv11 = np.array([l for _, _, l in pts], np.int32)
    cs11, it11 = oracle.assign_features_to_grid(xy11, 640.0, 480.0)
    q = lambda w, l0, l1: oracle.list_features_in_area_grid(xy11, lv11, cs11, it11, 640.0, 480.0, 190.0, 80.0, w, l0, l1)
    assert len(q(10, 0, 2)) == 1 and len(q(10, 1, 2)) == 0 and sorted(q(100, 0, 2).tolist()) == [0, 4]


def test_sincos_short_matches_glibc_for_every_float(tmp_path):
    """k_describe.hip computes the descriptor rotation's cos / sin with a short double sequence (Cody-Waite reduction +
    fdlibm kernels) instead of the device libm.  tools/sincos_check.c restates that sequence in C, operation for operation,
    and compares its result ROUNDED TO FLOAT with glibc's -- what the oracle uses -- for every float in [0, 6.4]
    (1,087,163,598 values, every angle the extractor can produce): no mismatch."""
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    exe = tmp_path / "sincos_check"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", str(root / "tools" / "sincos_check.c"), "-o", str(exe), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600).stdout
    assert "floats checked: 1087163598  sin mismatches 0  cos mismatches 0" in out, out
