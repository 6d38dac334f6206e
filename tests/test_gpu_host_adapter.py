"""GPU test of the C++ adapters (orb_slam3_amd/host/*.h: HipORBFeatureExtractor, HipSNNMatcher,
HipOrientationValidator) driven like the reference's callers by tests/host/adapter_main.cpp, which is prebuilt
against stand-ins of the reference headers (tests/host/stubs: this image has no Eigen)."""
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

from orb_slam3_amd import synth

ROOT = Path(__file__).resolve().parents[1]
BIN = ROOT / "tests" / "host" / "adapter_test.bin"


def test_adapter_sources_cite_every_virtual():
    """CPU: the adapter overrides all 11 pure virtuals of IFeatureExtractor (ifeature_extractor.h:33-53)."""
    src = (ROOT / "orb_slam3_amd" / "host" / "hip_orb_feature_extractor.h").read_text()
    for name in ("Extract", "Type", "GetAcceptableSquareError", "ComputeInvariantDistances", "PredictScale",
                 "ComputeDistance", "GetScaleFactors", "GetHighThreshold", "GetLowThreshold", "Serialize"):
        assert f"{name}(" in src and "override" in src


@pytest.mark.gpu
def test_cpp_adapter_end_to_end(oracle, tmp_path):
    assert BIN.exists(), "run python __graft_entry__.py (build) first"
    W, H, NF = 320, 240, 500
    frames = synth.sequence(5, 2, H, W)
    for i in range(2):
        frames[i].tofile(tmp_path / f"img{i}.raw")
    out = tmp_path / "out.bin"
    subprocess.run([str(BIN), str(W), str(H), str(NF), str(tmp_path / "img0.raw"), str(tmp_path / "img1.raw"), str(out)],
                   check=True, timeout=120)
    buf = out.read_bytes()
    pos = 0

    def take(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, buf, pos)
        pos += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    p = oracle.params(W, H)
    feats = []
    for i in range(2):
        n = take("i")
        rec = np.frombuffer(buf, dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("angle", "<f8"), ("size", "<f8"), ("level", "<i4")]),
                            count=n, offset=pos)
        pos += n * 36
        desc = np.frombuffer(buf, np.uint8, n * 32, pos).reshape(n, 32)
        pos += n * 32
        ok, od = oracle.extract(p, frames[i], NF)
        assert n == len(ok)
        for fld in ("x", "y", "angle", "size", "level"):
            assert np.array_equal(rec[fld], ok[fld]), fld
        assert np.array_equal(desc, od)
        feats.append((ok, od))
    n_before, discarded, n_after = take("i"), take("i"), take("i")
    got = dict(take("ii") for _ in range(n_after))
    (k0, d0), (k1, d1) = feats
    mf, _ = oracle.match_area(k1, d1, k0, d0, 100, 50, 0.9)
    assert n_before == (mf >= 0).sum()
    mf, nd = oracle.orientation_validate(k1["angle"], k0["angle"], mf)
    assert discarded == nd and got == {i: int(j) for i, j in enumerate(mf) if j >= 0}
    n_proj = take("i")
    proj = dict(take("ii") for _ in range(n_proj))
    sel = np.nonzero(k0["level"] <= 1)[0]
    sf = [1.0]
    for _ in range(7):
        sf.append(sf[-1] * 1.2)
    lv = k0["level"][sel]
    win = np.array([int(15.7 * sf[l]) for l in lv], np.uint64)
    omf, _ = oracle.snn_match(d0[sel], d1, 100, 0.9, to_xy=np.stack([k0["x"][sel], k0["y"][sel]], 1), to_window=win,
                              to_min_level=np.maximum(lv - 1, 0), to_max_level=lv + 1,
                              from_xy=np.stack([k1["x"], k1["y"]], 1), from_level=k1["level"])
    assert proj == {int(sel[i]): int(j) for i, j in enumerate(omf) if j >= 0} and len(proj) > 20
    n_bow = take("i")
    bow = dict(take("ii") for _ in range(n_bow))
    fv0, fv1 = {}, {}
    for j in range(len(d0)):
        fv0.setdefault(int(d0[j, 0]) % 16, []).append(j)
    for j in range(len(d1)):
        fv1.setdefault(int(d1[j, 0]) % 16, []).append(j)
    omb, _ = oracle.match_bow(d1, d0, fv1, fv0, 100, 0.9)
    assert bow == {i: int(j) for i, j in enumerate(omb) if j >= 0} and len(bow) > 5
    # HipBowVocabulary::transform on the k = 4, L = 3 vocabulary adapter_main.cpp builds from frame 0's descriptors
    k, L = 4, 3
    n_nodes = sum(k ** l for l in range(L + 1))
    first_leaf = n_nodes - k ** L
    co, ch, nxt = [0], [], 1
    for v in range(n_nodes):
        if v < first_leaf:
            ch.extend(range(nxt, nxt + k))
            nxt += k
        co.append(len(ch))
    ndesc = np.stack([d0[v % len(d0)] for v in range(n_nodes)])
    wid = np.array([v - first_leaf if v >= first_leaf else 0 for v in range(n_nodes)], np.uint32)
    wt = [0.0 if v % 7 == 3 else 0.5 + 0.25 * (v % 5) for v in range(n_nodes)]
    ow, on = oracle.bow_transform(d1, co, ch, ndesc, wid, L, 2)
    v, fv = {}, {}
    for i in range(len(d1)):
        w = wt[first_leaf + int(ow[i])]
        if w > 0:
            v[int(ow[i])] = v.get(int(ow[i]), 0.0) + w
            fv.setdefault(int(on[i]), []).append(i)
    tot = 0.0
    for kk in sorted(v):
        tot += abs(v[kk])
    n_bv = take("i")
    got_bv = dict(take("id") for _ in range(n_bv))
    assert got_bv == {kk: v[kk] / tot for kk in sorted(v)} and n_bv > 3
    n_fv = take("i")
    got_fv = {}
    for _ in range(n_fv):
        node, cnt = take("ii")
        got_fv[node] = [take("i") for _ in range(cnt)]
    assert got_fv == fv
    n_groups = take("i")
    got_best = [take("i") for _ in range(n_groups)]
    grp = [d0[g::7] for g in range(7)] + [np.zeros((0, 32), np.uint8)]
    off = np.cumsum([0] + [len(g) for g in grp]).astype(np.int32)
    assert got_best == oracle.distinctive_descriptor(np.concatenate(grp), off).tolist() and got_best[-1] == -1
    n_geom = take("i")
    assert n_geom == len(k1)
    rec = np.frombuffer(buf, np.dtype([("ok", "u1"), ("und", "<f8", 2), ("unp", "<f8", 3)]), n_geom, pos)
    pos += n_geom * 41
    wund, wunp, wok = oracle.keypoint_geometry(k1, oracle.camera(258.6, 258.2, 160.2, 120.1, 0, (-0.28, 0.07, 2e-4, 1.8e-5, 0.0)))
    assert np.array_equal(rec["ok"], wok) and np.array_equal(rec["und"], wund) and np.array_equal(rec["unp"], wunp)
    # HipKeyFrameIndex: the database adapter_main.cpp builds from frame 0's descriptors
    kf_bow = []
    for g in range(9):
        bow = {}
        for j in range(g, len(d0), 9):
            w = (j * 13) % 97
            bow[w] = bow.get(w, 0.0) + (1.0 + float(d0[j, 1]))
        tot = 0.0
        for w in sorted(bow):
            tot += abs(bow[w])
        kf_bow.append({w: bow[w] / tot for w in sorted(bow)})
    live = [g for g in range(9) if g != 4]
    offs = np.cumsum([0] + [len(kf_bow[g]) for g in live]).astype(np.int64)
    words = np.concatenate([np.fromiter(kf_bow[g].keys(), np.uint32) for g in live])
    vals = np.concatenate([np.fromiter(kf_bow[g].values(), np.float64) for g in live])
    q = kf_bow[6]
    sh, sc = oracle.bow_score(np.fromiter(q.keys(), np.uint32), np.fromiter(q.values(), np.float64), offs, words, vals, 0)
    n_hits = take("i")
    got_hits = {}
    for _ in range(n_hits):
        kid, shw, score = take("QQd")
        got_hits[kid] = (shw, score)
    assert got_hits == {100 + g: (int(sh[n]), float(sc[n])) for n, g in enumerate(live)}
    assert take("I") == oracle.predict_scale(1.0, 3.0, 1.2, 8)
    s3 = 1.0 * 1.2 * 1.2 * 1.2
    assert take("d") == s3 * s3
    assert take("I") == oracle.hamming256(d0[0], d0[1])
    assert take("i") == 4 + 4 + 8 + 8 + 4 + 4  # Serialize: 2 unsigned, double, size_t, 2 unsigned (:875-885)
