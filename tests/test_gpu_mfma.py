"""The exploratory i8-MFMA engine of the brute-force matcher (orbgpu_set_bruteforce_engine) against the oracle and against
the popcount engine: same indices and distances bit for bit, including ties (duplicate rows), contested claims, partial
tiles, chunk boundaries and thresholds that keep every / no candidate.  Reference semantics:
second_nearest_neighbor_matcher.hpp:40-71."""
import numpy as np
import pytest

from orb_slam3_amd import synth

pytestmark = pytest.mark.gpu


def _run(lib, fe, dt, df, n_to, n_from, th, ratio, di, dd, engine):
    assert lib.orbgpu_set_bruteforce_engine(fe.handle, engine) == 0
    rc = lib.orbgpu_match_bruteforce_device(fe.handle, dt.data_ptr(), n_to, df.data_ptr(), n_from, th, ratio,
                                            di.data_ptr(), dd.data_ptr())
    assert rc == 0, lib.orbgpu_last_error(fe.handle)
    assert lib.orbgpu_synchronize(fe.handle) == 0
    return di.cpu().numpy().copy(), dd.cpu().numpy().view(np.uint32).copy()


@pytest.mark.parametrize("n_to,n_from", [(1, 1), (31, 33), (513, 257), (3000, 3000), (700, 20011), (20011, 700),
                                         (40000, 9000)])
def test_mfma_engine_parity(oracle, n_to, n_from):
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    rng = np.random.default_rng(n_to * 7 + n_from)
    frm = synth.descriptors(900 + n_from, n_from)
    to = synth.descriptors(1000 + n_to, n_to, planted_from=frm, planted_frac=0.4, max_flips=70)
    if n_from > 40:
        dup = rng.choice(n_from - 1, 12, replace=False)
        frm[dup + 1] = frm[dup]                                   # best == second ties
        frm[n_from - 1] = frm[0]                                   # first row against the last (other chunk / lane half)
    if n_to > 600:
        src = rng.choice(n_to, 20, replace=False)
        for s in src:
            for off in (1, 31, 32, 33, 127, 128, 511, 512):       # other lane half, other tile, other wave, other block
                to[(s + off) % n_to] = to[s]
    to[0] = frm[n_from // 2]                                      # distance 0
    to[n_to - 1] = ~frm[n_from // 2]                              # distance 256
    fe = ORBFeatureExtractor(640, 480, 1.2, 8, 20, 7, max_features=1000)
    dev = torch.device("cuda", 0)
    dt, df = torch.from_numpy(to).to(dev), torch.from_numpy(frm).to(dev)
    di = torch.zeros(n_to, dtype=torch.int32, device=dev)
    dd = torch.zeros(n_to, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    settings = ((50, 0.9), (100, 0.8), (256, 1.0), (0, 0.9), (128, 0.6)) if n_to * n_from <= 2e7 else ((100, 0.8), (50, 0.9))
    for th, ratio in settings:
        mi, md = _run(lib, fe, dt, df, n_to, n_from, th, ratio, di, dd, _lib.ENGINE_MFMA_I8)
        pi, pd = _run(lib, fe, dt, df, n_to, n_from, th, ratio, di, dd, _lib.ENGINE_POPCOUNT)
        omf, omd = oracle.snn_match(to, frm, th, ratio)
        assert np.array_equal(mi, omf), (th, ratio)
        assert np.array_equal(pi, omf), (th, ratio)
        sel = omf >= 0
        assert np.array_equal(md[sel], omd[sel]) and np.array_equal(pd[sel], omd[sel])
    fe.close()
