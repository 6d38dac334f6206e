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


def test_noalloc_entry_point_allocates_nothing():
    """A fresh context whose growable match scratch is still at its initial 1 MiB: the no-alloc entry point runs FIRST, on
    a problem that needs several MiB of scratch, between two hipMemGetInfo readings that must agree (a call routed through
    the context's growable buffer would have had to hipMalloc); only then the allocating entry point, for the comparison
    (and that one does take device memory)."""
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n_to, n_from = 30000, 400000            # claims 3.2 MB + partials: far above the context's initial 1 MiB
    frm = synth.descriptors(77, n_from)
    to = synth.descriptors(78, n_to, planted_from=frm[:5000], planted_frac=0.4, max_flips=50)
    dt, df = torch.from_numpy(to).to(dev), torch.from_numpy(frm).to(dev)
    outs = [(torch.full((n_to,), 7, dtype=torch.int32, device=dev), torch.full((n_to,), 7, dtype=torch.int32, device=dev))
            for _ in range(2)]
    fe = ORBFeatureExtractor(640, 480, 1.2, 8, 20, 7, max_features=1000)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    rc = lib.orbgpu_match_bruteforce_noalloc_device(fe.handle, dt.data_ptr(), n_to, df.data_ptr(), n_from, 50, 0.9,
                                                    outs[0][0].data_ptr(), outs[0][1].data_ptr())
    assert rc == 0, lib.orbgpu_last_error(fe.handle)
    assert lib.orbgpu_synchronize(fe.handle) == 0
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free1 == free0, "the no-alloc entry point changed the device's free memory"
    rc = lib.orbgpu_match_bruteforce_device(fe.handle, dt.data_ptr(), n_to, df.data_ptr(), n_from, 50, 0.9,
                                            outs[1][0].data_ptr(), outs[1][1].data_ptr())
    assert rc == 0, lib.orbgpu_last_error(fe.handle)
    assert lib.orbgpu_synchronize(fe.handle) == 0
    free2, _ = torch.cuda.mem_get_info(dev)
    assert free2 < free1, "the allocating entry point was expected to grow the context's scratch"
    gi, ri = outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy()
    assert np.array_equal(gi, ri) and (ri >= 0).sum() > 1000
    sel = ri >= 0
    assert np.array_equal(outs[0][1].cpu().numpy()[sel], outs[1][1].cpu().numpy()[sel])
    fe.close()
