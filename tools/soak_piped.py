"""Soak: many orbgpu_extract_batch_device + link calls on ONE context under orbgpu_set_input_ready (two working sets, the
head of a call beside the previous call's tail), three different 256-frame batches rotating so that the two calls that
share a working set never hold the same images; every call's outputs against the plain-order answers.
Usage: python tools/soak_piped.py [CALLS]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from orb_slam3_amd import _lib, synth
from orb_slam3_amd.features import ORBFeatureExtractor

lib = _lib.load()
W, H, NF, B = 640, 480, 1000, 256
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda", 0)
base = synth.sequence(4321, 96, H, W)
imgs = [torch.from_numpy(np.ascontiguousarray(np.concatenate([base[32 * i:32 * (i + 1)]] * 8)[:B])).to(dev) for i in range(3)]
fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, device=0, max_batch=B, max_features=NF)
cap = fe.geometry(NF).keypoint_capacity
h = fe.handle
to = torch.arange(1, B, dtype=torch.int32, device=dev)
frm = torch.arange(0, B - 1, dtype=torch.int32, device=dev)


def bufs():
    return (torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
            torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros((B - 1, cap), dtype=torch.int32, device=dev),
            torch.zeros((B - 1, cap), dtype=torch.int32, device=dev), torch.zeros(B - 1, dtype=torch.int32, device=dev))


def step(img, o):
    kp, desc, n, mi, md, mc = o
    assert lib.orbgpu_extract_batch_device(h, img.data_ptr(), W * H, W, B, NF, kp.data_ptr(), desc.data_ptr(), cap, n.data_ptr()) == 0
    assert lib.orbgpu_match_frames_device(h, kp.data_ptr(), desc.data_ptr(), n.data_ptr(), cap, to.data_ptr(), frm.data_ptr(), B - 1,
                                          100, 50, 0.9, 1, mi.data_ptr(), md.data_ptr(), mc.data_ptr()) == 0


def same(a, b):
    if not (torch.equal(a[2], b[2]) and torch.equal(a[5], b[5])):
        return False
    n = a[2].cpu().numpy()
    m = a[5].cpu().numpy()
    ad, bd, ak, bk = a[1].cpu().numpy(), b[1].cpu().numpy(), a[0].cpu().numpy(), b[0].cpu().numpy()
    am, bm = a[3].cpu().numpy(), b[3].cpu().numpy()
    return all(np.array_equal(ad[f, :n[f]], bd[f, :n[f]]) and np.array_equal(ak[f, :n[f]], bk[f, :n[f]]) for f in range(B)) and \
        all(np.array_equal(am[p, :m[p]], bm[p, :m[p]]) for p in range(B - 1))


ref = [bufs() for _ in range(3)]
for i in range(3):
    step(imgs[i], ref[i])
assert lib.orbgpu_synchronize(h) == 0
assert lib.orbgpu_set_input_ready(h, 1, None) == 0
ring = [bufs() for _ in range(6)]
bad = 0
for c0 in range(0, calls, 6):
    for j in range(6):
        step(imgs[(c0 + j) % 3], ring[j])
    assert lib.orbgpu_synchronize(h) == 0
    for j in range(6):
        if not same(ring[j], ref[(c0 + j) % 3]):
            bad += 1
            print("call", c0 + j, "DIFFERS", flush=True)
    if c0 % 30 == 0:
        print("..", c0, flush=True)
print("done:", calls, "calls,", bad, "differ")
