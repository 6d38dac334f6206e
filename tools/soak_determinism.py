"""Soak: the same device batch extracted and linked many times; every repetition must reproduce the first one bit for
bit (a missing barrier or an order-dependent atomic would show up as a rare difference).
Usage: python tools/soak_determinism.py [REPS] [W H FEATURES FRAMES]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from orb_slam3_amd import _lib, synth  # noqa: E402
from orb_slam3_amd.features import ORBFeatureExtractor  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W, H, NF, B = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (640, 480, 1000, 256)
lib = _lib.load()
dev = torch.device("cuda", 0)
base = synth.sequence(4242, min(B, 32), H, W)
frames = np.concatenate([base] * ((B + len(base) - 1) // len(base)))[:B]
fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
cap = fe.geometry(NF).keypoint_capacity
d_imgs = torch.from_numpy(np.ascontiguousarray(frames)).to(dev)
to = torch.arange(1, B, dtype=torch.int32, device=dev)
frm = torch.arange(0, B - 1, dtype=torch.int32, device=dev)
ref = None
bad = 0
for r in range(reps):
    d_kp = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_m = torch.zeros((B - 1, cap), dtype=torch.int32, device=dev)
    d_d = torch.zeros((B - 1, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros(B - 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    assert lib.orbgpu_extract_batch_device(fe.handle, d_imgs.data_ptr(), W * H, W, B, NF, d_kp.data_ptr(), d_desc.data_ptr(), cap,
                                           d_n.data_ptr()) == 0
    assert lib.orbgpu_match_frames_device(fe.handle, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, to.data_ptr(), frm.data_ptr(),
                                          B - 1, 100, 50, 0.9, 1, d_m.data_ptr(), d_d.data_ptr(), d_nm.data_ptr()) == 0
    assert lib.orbgpu_synchronize(fe.handle) == 0
    cur = (d_kp, d_desc, d_n, d_m, d_nm)
    if ref is None:
        ref = cur
        print("reference:", int(d_n.sum().item()), "keypoints,", int(d_nm.sum().item()), "links", flush=True)
    else:
        same = all(torch.equal(a, b) for a, b in zip(cur, ref))
        if not same:
            bad += 1
            print("DIFFERENCE in repetition", r, [bool(torch.equal(a, b)) for a, b in zip(cur, ref)], flush=True)
    if r % 50 == 49:
        print("..", r + 1, flush=True)
print("done:", reps, "repetitions,", bad, "differ")
