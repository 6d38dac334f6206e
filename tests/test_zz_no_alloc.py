"""orbgpu_match_bruteforce_noalloc_device -- the entry point that allocates nothing (its scratch is a fixed __device__
array inside liborbgpu.so) -- against orbgpu_match_bruteforce_device on the same inputs, bit for bit, on two calls in a
row, the second with a smaller input (the array keeps the first call's contents).  Collected last on purpose."""
import numpy as np
import pytest

from orb_slam3_amd import synth

pytestmark = pytest.mark.gpu


def _both(lib, fe, torch, to, frm, th, ratio):
    dev = torch.device("cuda", 0)
    n_to, n_from = len(to), len(frm)
    dt, df = torch.from_numpy(to).to(dev), torch.from_numpy(frm).to(dev)
    out = []
    for fn in (lib.orbgpu_match_bruteforce_device, lib.orbgpu_match_bruteforce_noalloc_device):
        di = torch.full((n_to,), 12345, dtype=torch.int32, device=dev)
        dd = torch.full((n_to,), 12345, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        rc = fn(fe.handle, dt.data_ptr(), n_to, df.data_ptr(), n_from, th, ratio, di.data_ptr(), dd.data_ptr())
        assert rc == 0, lib.orbgpu_last_error(fe.handle)
        assert lib.orbgpu_synchronize(fe.handle) == 0
        out.append((di.cpu().numpy(), dd.cpu().numpy()))
    return out


def test_noalloc_bruteforce_matches_allocating_entry_point(oracle):
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    fe = ORBFeatureExtractor(640, 480, 1.2, 8, 20, 7, max_features=1000)
    # first call: large enough that the fixed scratch limits the number of 'from' chunks; second: small, and with
    # fewer rows than the first call left behind in the array
    # (20000 x 1,000,000: the claims take 8.0 of the 8 MiB, which leaves room for 2 'from' chunks where the allocating entry
    # point uses 27; too many pairs for the CPU oracle, so that case is checked against the allocating entry point only)
    for n_to, n_from, th, ratio in ((20000, 1_000_000, 50, 0.9), (40000, 9000, 50, 0.9), (700, 300, 100, 0.8), (1, 1, 50, 0.9)):
        frm = synth.descriptors(900 + n_from, n_from)
        to = synth.descriptors(901 + n_to, n_to, planted_from=frm, planted_frac=0.5, max_flips=60)
        if n_to > 600:
            to[5] = to[4]          # contested claim
            to[600] = to[4]
            frm[11] = frm[10]      # best == second
        (ri, rd), (gi, gd) = _both(lib, fe, torch, to, frm, th, ratio)
        assert np.array_equal(gi, ri)
        sel = ri >= 0
        assert np.array_equal(gd[sel], rd[sel])
        if n_to * n_from <= 4e8:
            omf, _ = oracle.snn_match(to, frm, th, ratio)
            assert np.array_equal(gi, omf)
        assert n_to == 1 or sel.sum() > 50
    # empty sides
    import ctypes as C
    dev = torch.device("cuda", 0)
    di = torch.zeros(4, dtype=torch.int32, device=dev)
    dd = torch.zeros(4, dtype=torch.int32, device=dev)
    dt = torch.zeros((4, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    assert lib.orbgpu_match_bruteforce_noalloc_device(fe.handle, dt.data_ptr(), 4, None, 0, 50, 0.9, di.data_ptr(), dd.data_ptr()) == 0
    assert lib.orbgpu_synchronize(fe.handle) == 0 and (di.cpu().numpy() == -1).all()
    assert lib.orbgpu_match_bruteforce_noalloc_device(fe.handle, None, 0, dt.data_ptr(), 4, 50, 0.9, None, None) == 0
    # beyond the fixed scratch: rejected with a clear error, nothing launched
    rc = lib.orbgpu_match_bruteforce_noalloc_device(fe.handle, dt.data_ptr(), 2_000_000, dt.data_ptr(), 4, 50, 0.9,
                                                    di.data_ptr(), dd.data_ptr())
    assert rc == _lib.E_CAPACITY and b"MiB scratch" in lib.orbgpu_last_error(fe.handle)
    rc = lib.orbgpu_match_bruteforce_noalloc_device(fe.handle, dt.data_ptr(), 4, dt.data_ptr(), 1_100_000, 50, 0.9,
                                                    di.data_ptr(), dd.data_ptr())
    assert rc == _lib.E_CAPACITY
    fe.close()
