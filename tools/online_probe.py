"""Online call shape (one frame / one search per call, host to host): wall time of orbgpu_extract and orbgpu_match_window,
and -- run under `rocprofv3 --kernel-trace` -- the kernel timeline of the last single-frame extraction.
Usage: python tools/online_probe.py [n_calls]"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401  (one HIP runtime for the process)
from orb_slam3_amd import _lib, synth
from orb_slam3_amd.features import ORBFeatureExtractor, Features

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W, H, NF = 640, 480, 1000
lib = _lib.load()
fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=1, max_features=NF)
frames = synth.sequence(1235, 2, H, W)
cap = fe.geometry(NF).keypoint_capacity
kp = np.zeros((2, cap), _lib.KP_DTYPE)
desc = np.zeros((2, cap, 32), np.uint8)
n = C.c_int32(0)
cnt = [0, 0]
t_ext = []
for it in range(n_calls):
    f = it & 1
    t0 = time.perf_counter()
    rc = lib.orbgpu_extract(fe.handle, frames[f].ctypes.data, W, H, W, NF, kp[f].ctypes.data, desc[f].ctypes.data, cap, C.byref(n))
    t_ext.append(time.perf_counter() - t0)
    assert rc == 0, lib.orbgpu_last_error(fe.handle)
    cnt[f] = n.value
g = fe.geometry(NF)
sc = np.array([g.scale[i] for i in range(g.levels)])
k0, k1 = kp[0, :cnt[0]], kp[1, :cnt[1]]
to_xy, from_xy = np.stack([k0["x"], k0["y"]], 1).copy(), np.stack([k1["x"], k1["y"]], 1).copy()
win = (15.0 * sc[k0["level"]]).astype(np.uint64)
lmin, lmax = np.maximum(k0["level"] - 1, 0).astype(np.int32), (k0["level"] + 1).astype(np.int32)
flv = k1["level"].astype(np.int32).copy()
td, fd = desc[0, :cnt[0]].copy(), desc[1, :cnt[1]].copy()
a = _lib.WindowArgs(_lib.ptr(td), _lib.ptr(to_xy), _lib.ptr(win), _lib.ptr(lmin), _lib.ptr(lmax), None, cnt[0],
                    _lib.ptr(fd), _lib.ptr(from_xy), _lib.ptr(flv), cnt[1], 100, 0.9)
idx, dist = np.zeros(cnt[0], np.int32), np.zeros(cnt[0], np.uint32)
t_win = []
for it in range(n_calls):
    t0 = time.perf_counter()
    rc = lib.orbgpu_match_window(fe.handle, C.byref(a), _lib.ptr(idx), _lib.ptr(dist))
    t_win.append(time.perf_counter() - t0)
    assert rc == 0
med = lambda v: float(np.median(v[len(v) // 4:])) * 1e6
print(f"orbgpu_extract {W}x{H}: {med(t_ext):.1f} us host to host (min {min(t_ext) * 1e6:.1f}); "
      f"orbgpu_match_window {cnt[0]} x {cnt[1]}: {med(t_win):.1f} us (min {min(t_win) * 1e6:.1f}), {(idx >= 0).sum()} matches")
fe.close()
