"""Experiment: stage times of the 256-frame batch with minThFAST = 7 (default) and = iniThFAST = 20.
The second run bounds what a two-threshold FAST kernel could save (its output differs, by construction)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from orb_slam3_amd import _lib, synth
from orb_slam3_amd.features import ORBFeatureExtractor

lib = _lib.load()
W, H, NF, B = 640, 480, 1000, 256
base = synth.sequence(1235, 32, H, W)
frames = np.concatenate([base] * 8)[:B]
dev = torch.device("cuda", 0)
d_imgs = torch.from_numpy(np.ascontiguousarray(frames)).to(dev)
for tmin in (7, 20, 7, 20):
    fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, tmin, device=0, max_batch=B, max_features=NF)
    cap = fe.geometry(NF).keypoint_capacity
    d_kp = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for it in range(8):
        rc = lib.orbgpu_extract_batch_device(fe.handle, d_imgs.data_ptr(), W * H, W, B, NF, d_kp.data_ptr(),
                                             d_desc.data_ptr(), cap, d_n.data_ptr())
        assert rc == 0
        if it == 2:
            lib.orbgpu_synchronize(fe.handle)
    lib.orbgpu_synchronize(fe.handle)
    st = fe.stage_times()
    print(tmin, {k: round(v, 4) for k, v in st.items()}, "kp mean", float(d_n.float().mean()), flush=True)
    fe.close()
