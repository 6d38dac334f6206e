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
    for l, (a, b) in enumerate(zip(_levels(d["score"], g), _levels(ex["score"], g))):
        assert np.array_equal(a, b), f"FAST score level {l}"
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
