"""Committed golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the oracle):
the CPU test pins the oracle against them; the GPU test compares the HIP path with the fixtures alone."""
from pathlib import Path

import numpy as np
import pytest

GOLD = sorted((Path(__file__).resolve().parent / "golden").glob("*.npz"))


def _d(idx):
    return {i: int(j) for i, j in enumerate(idx) if j >= 0}


@pytest.mark.parametrize("path", GOLD, ids=lambda p: p.stem)
def test_oracle_reproduces_golden(oracle, path):
    z = np.load(path)
    w, h, levels, nfeat = (int(v) for v in z["params"])
    p = oracle.params(w, h, levels=levels)
    k0, d0, ex = oracle.extract(p, z["frames"][0], nfeat, dump=True)
    k1, d1 = oracle.extract(p, z["frames"][1], nfeat)
    assert np.array_equal(k0, z["kp0"]) and np.array_equal(d0, z["desc0"])
    assert np.array_equal(k1, z["kp1"]) and np.array_equal(d1, z["desc1"])
    assert np.array_equal(ex["pyramid"], z["pyramid0"]) and np.array_equal(ex["score"], z["score0"])
    mf, md = oracle.match_area(k1, d1, k0, d0, 100, 50, 0.9)
    assert np.array_equal(mf, z["link"])
    mfo, nd = oracle.orientation_validate(k1["angle"], k0["angle"], mf)
    assert np.array_equal(mfo, z["link_filtered"]) and nd == int(z["n_discarded"])
    bf, _ = oracle.snn_match(d1, d0, 100, 0.8)
    assert np.array_equal(bf, z["brute"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=lambda p: p.stem)
def test_gpu_reproduces_golden(path):
    from orb_slam3_amd.features import ORBFeatureExtractor, SNNMatcher, OrientationValidator
    z = np.load(path)
    w, h, levels, nfeat = (int(v) for v in z["params"])
    fe = ORBFeatureExtractor(w, h, 1.2, levels, 20, 7, max_batch=2, max_features=nfeat)
    f0, f1 = fe.ExtractBatch([z["frames"][0], z["frames"][1]], nfeat)
    for f, k, d in ((f0, z["kp0"], z["desc0"]), (f1, z["kp1"], z["desc1"])):
        assert np.array_equal(f.keypoints["x"], k["x"]) and np.array_equal(f.keypoints["y"], k["y"])
        assert np.array_equal(f.keypoints["level"], k["level"])
        assert np.array_equal(f.keypoints["angle"], k["angle"].astype(np.float32))
        assert np.array_equal(f.descriptors, d)
    dump = fe.debug_dump(0)
    assert np.array_equal(dump["pyramid"], z["pyramid0"])
    # the device fills the score map only where the cells read it: [19, w-20] x [19, h-20] of every level
    g, off = fe.geometry(nfeat), 0
    for l in range(g.levels):
        n = g.w[l] * g.h[l]
        a = dump["score"][off:off + n].reshape(g.h[l], g.w[l])
        b = z["score0"][off:off + n].reshape(g.h[l], g.w[l])
        assert np.array_equal(a[19:-19, 19:-19], b[19:-19, 19:-19]), f"score level {l}"
        off += n
    assert dump["n_cands"] == z["n_cands0"].tolist()
    got = SNNMatcher(0.9, 50, fe).MatchArea(f1, f0, 100)
    assert got == _d(z["link"])
    nd = OrientationValidator(f1.keypoints, f0.keypoints, fe).Validate(got)
    assert nd == int(z["n_discarded"]) and got == _d(z["link_filtered"])
    assert SNNMatcher(0.8, 100, fe).MatchBruteForce(f1.descriptors, f0.descriptors) == _d(z["brute"])
    fe.close()
