"""Times orbgpu_match_window_segments_device against orbgpu_match_window_grid_segments_device on synthetic keypoint sets
(uniform positions, planted matches).  Usage: python tools/time_grid_match.py W H n_kp n_pairs radius"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from orb_slam3_amd import _lib  # noqa: E402
from orb_slam3_amd.features import ORBFeatureExtractor  # noqa: E402


def main():
    W, H, n, S, radius = (float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
    lib = _lib.load()
    fe = ORBFeatureExtractor(640, 480, 1.2, 8, 20, 7, max_features=1000)
    rng = np.random.default_rng(3)
    N = S * n
    from_desc = rng.integers(0, 256, (N, 32), dtype=np.uint8)
    to_desc = from_desc.copy()
    to_desc[:, :2] ^= rng.integers(0, 256, (N, 2)).astype(np.uint8)
    from_xy = rng.uniform((0, 0), (W, H), (N, 2))
    to_xy = from_xy + rng.uniform(-4, 4, (N, 2))
    lv = np.minimum(rng.geometric(0.3, N) - 1, 7).astype(np.int32)
    win = (np.float64(radius) * np.float64(1.2) ** lv).astype(np.uint64)
    lmin, lmax = np.maximum(lv - 1, 0).astype(np.int32), (lv + 1).astype(np.int32)
    segs = np.zeros(S, _lib.SEGMENT_DTYPE)
    for s in range(S):
        segs[s] = (s * n, n, s * n, n)
    dev = torch.device("cuda", 0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in dict(
        to_desc=to_desc, from_desc=from_desc, to_xy=to_xy, from_xy=from_xy, win=win.view(np.int64), lmin=lmin, lmax=lmax,
        lv=lv, segs=segs.view(np.int32).reshape(S, 4)).items()}
    o = [torch.zeros(N, dtype=torch.int32, device=dev) for _ in range(4)]
    a = _lib.WindowArgs(t["to_desc"].data_ptr(), t["to_xy"].data_ptr(), t["win"].data_ptr(), t["lmin"].data_ptr(),
                        t["lmax"].data_ptr(), None, N, t["from_desc"].data_ptr(), t["from_xy"].data_ptr(),
                        t["lv"].data_ptr(), N, 100, 0.9)
    torch.cuda.synchronize()

    def plain():
        assert lib.orbgpu_match_window_segments_device(fe.handle, C.byref(a), t["segs"].data_ptr(), S, n, o[0].data_ptr(),
                                                       o[1].data_ptr()) == 0

    def grid():
        assert lib.orbgpu_match_window_grid_segments_device(fe.handle, C.byref(a), t["segs"].data_ptr(), S, n, n, W, H,
                                                            o[2].data_ptr(), o[3].data_ptr()) == 0
    res = {}
    for name, fn in (("plain", plain), ("grid", grid)):
        for _ in range(3):
            fn()
        lib.orbgpu_synchronize(fe.handle)
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        lib.orbgpu_synchronize(fe.handle)
        res[name] = (time.perf_counter() - t0) / 10 * 1e3
    same = bool(torch.equal(o[0], o[2]) and torch.equal(o[1], o[3]))
    print(f"{int(W)}x{int(H)} n={n} pairs={S} radius={radius}: plain {res['plain']:.3f} ms, grid {res['grid']:.3f} ms, "
          f"identical={same}, matched={(o[2] >= 0).sum().item()}/{N}")
    assert same


main()
