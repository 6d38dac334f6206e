"""Probe: K extract + link steps on one context, plain and with orbgpu_set_input_ready(ctx, 1, NULL) (the head of a call
beside the previous call's tail).  Usage: python tools/piped_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from orb_slam3_amd import _lib, synth
from orb_slam3_amd.features import ORBFeatureExtractor

lib = _lib.load()
W, H, NF, B = 640, 480, 1000, 256
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
base = synth.sequence(1235, 32, H, W)
frames = np.concatenate([base] * 8)[:B]
dev = torch.device("cuda", 0)
d_imgs = torch.from_numpy(np.ascontiguousarray(frames)).to(dev)
# PROBE_IDLE_CONTEXTS=n: n more contexts created (and used once) first, to see what streams that merely exist cost
idle = []
for _ in range(int(os.environ.get("PROBE_IDLE_CONTEXTS", "0"))):
    fx = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, device=0, max_batch=B, max_features=NF)
    cx = fx.geometry(NF).keypoint_capacity
    ox = (torch.zeros((B, cx, 32), dtype=torch.uint8, device=dev), torch.zeros((B, cx, 32), dtype=torch.uint8, device=dev),
          torch.zeros(B, dtype=torch.int32, device=dev))
    assert lib.orbgpu_extract_batch_device(fx.handle, d_imgs.data_ptr(), W * H, W, B, NF, ox[0].data_ptr(), ox[1].data_ptr(), cx, ox[2].data_ptr()) == 0
    assert lib.orbgpu_synchronize(fx.handle) == 0
    idle.append(fx)
fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, device=0, max_batch=B, max_features=NF)
cap = fe.geometry(NF).keypoint_capacity
h = fe.handle


def bufs():
    return (torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
            torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros((B - 1, cap), dtype=torch.int32, device=dev),
            torch.zeros((B - 1, cap), dtype=torch.int32, device=dev), torch.zeros(B - 1, dtype=torch.int32, device=dev))


to = torch.arange(1, B, dtype=torch.int32, device=dev)
frm = torch.arange(0, B - 1, dtype=torch.int32, device=dev)


def step(o):
    kp, desc, n, mi, md, mc = o
    assert lib.orbgpu_extract_batch_device(h, d_imgs.data_ptr(), W * H, W, B, NF, kp.data_ptr(), desc.data_ptr(), cap, n.data_ptr()) == 0
    assert lib.orbgpu_match_frames_device(h, kp.data_ptr(), desc.data_ptr(), n.data_ptr(), cap, to.data_ptr(), frm.data_ptr(), B - 1,
                                          100, 50, 0.9, 1, mi.data_ptr(), md.data_ptr(), mc.data_ptr()) == 0


def run(o, k):
    assert lib.orbgpu_synchronize(h) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        step(o)
    assert lib.orbgpu_synchronize(h) == 0
    return (time.perf_counter() - t0) / k * 1e3


a, b = bufs(), bufs()
for _ in range(3):
    step(a)
for rep in range(3):
    assert lib.orbgpu_set_input_ready(h, 0, None) == 0
    t1 = run(a, K)
    assert lib.orbgpu_set_input_ready(h, 1, None) == 0
    t2 = run(b, K)
    print(f"plain {t1:.4f} ms/step ({B / t1:.1f} k frames/s)   input-ready {t2:.4f} ms/step ({B / t2:.1f} k frames/s)", flush=True)
for x, y in zip(a, b):
    assert torch.equal(x, y)
print("outputs identical")
