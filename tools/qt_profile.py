import sys, numpy as np, torch
sys.path.insert(0, ".")
from orb_slam3_amd import _lib, synth
from orb_slam3_amd.features import ORBFeatureExtractor
W, H, NF, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lib = _lib.load()
fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
cap = fe.geometry(NF).keypoint_capacity
frames = synth.sequence(1236, min(B, 4), H, W)
frames = np.ascontiguousarray(frames[np.arange(B) % len(frames)])
dev = torch.device("cuda", 0)
d_imgs = torch.from_numpy(frames).to(dev)
d_kp = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
d_n = torch.zeros(B, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for _ in range(2):
    assert lib.orbgpu_extract_batch_device(fe.handle, d_imgs.data_ptr(), W * H, W, B, NF, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr()) == 0
    assert lib.orbgpu_synchronize(fe.handle) == 0
