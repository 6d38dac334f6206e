"""Probe: K extract + link steps on ONE context against the same steps alternating between TWO contexts (each with its
own internal buffers and outputs), so that the tail of one batch (descriptors, link) can overlap the head of the next
(pyramid, FAST).  Usage: python tools/two_ctx_probe.py [steps]"""
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


class Ctx:
    def __init__(self):
        self.fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, device=0, max_batch=B, max_features=NF)
        self.cap = self.fe.geometry(NF).keypoint_capacity
        self.kp = torch.zeros((B, self.cap, 32), dtype=torch.uint8, device=dev)
        self.desc = torch.zeros((B, self.cap, 32), dtype=torch.uint8, device=dev)
        self.n = torch.zeros(B, dtype=torch.int32, device=dev)
        self.mi = torch.zeros((B - 1, self.cap), dtype=torch.int32, device=dev)
        self.md = torch.zeros((B - 1, self.cap), dtype=torch.int32, device=dev)
        self.mc = torch.zeros(B - 1, dtype=torch.int32, device=dev)
        self.to = torch.arange(1, B, dtype=torch.int32, device=dev)
        self.frm = torch.arange(0, B - 1, dtype=torch.int32, device=dev)

    def step(self):
        h = self.fe.handle
        assert lib.orbgpu_extract_batch_device(h, d_imgs.data_ptr(), W * H, W, B, NF, self.kp.data_ptr(), self.desc.data_ptr(),
                                               self.cap, self.n.data_ptr()) == 0
        assert lib.orbgpu_match_frames_device(h, self.kp.data_ptr(), self.desc.data_ptr(), self.n.data_ptr(), self.cap,
                                              self.to.data_ptr(), self.frm.data_ptr(), B - 1, 100, 50, 0.9, 1,
                                              self.mi.data_ptr(), self.md.data_ptr(), self.mc.data_ptr()) == 0

    def sync(self):
        assert lib.orbgpu_synchronize(self.fe.handle) == 0


def run(ctxs, k):
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        ctxs[i % len(ctxs)].step()
    for c in ctxs:
        c.sync()
    return (time.perf_counter() - t0) / k * 1e3


a, b = Ctx(), Ctx()
for c in (a, b):
    for _ in range(3):
        c.step()
    c.sync()
for rep in range(3):
    t1 = run([a], K)
    t2 = run([a, b], K)
    st1 = None
    a.fe.stage_times()
    t1 = run([a], K)
    st1 = a.fe.stage_times()["fast_score"]
    t2 = run([a, b], K)
    st2 = a.fe.stage_times()["fast_score"]
    print(f"   FAST launch of the levels >= 1: {st1:.4f} ms alone, {st2:.4f} ms with two contexts alternating")
    print(f"one context {t1:.4f} ms/step ({B / t1:.1f} k frames/s)   two contexts alternating {t2:.4f} ms/step ({B / t2:.1f} k frames/s)", flush=True)
ref_n = a.n.cpu().numpy().copy()
assert np.array_equal(ref_n, b.n.cpu().numpy()) and torch.equal(a.desc, b.desc) and torch.equal(a.mi, b.mi)
print("outputs of the two contexts identical")
