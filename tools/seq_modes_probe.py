"""Probe: the sequence runner over 20,000 resident (and pooled) frames with one context, one context under the input-ready
hint, and two contexts alternating.  Usage: python tools/seq_modes_probe.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from orb_slam3_amd.features import ORBFeatureExtractor
from orb_slam3_amd.sequence import PooledSource, ResidentSource, SequenceRunner
W, H, NF, B = 640, 480, 1000, 256
dev = torch.device("cuda", 0)
fe_a = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
pool = PooledSource(17, pool_frames=64, height=H, width=W)
res = ResidentSource(pool, dev, repeat=B // 64)
n = 20000
def t(runner, tag):
    runner.run(3 * B)
    best = 1e9
    for _ in range(3):
        r = runner.run(n)
        best = min(best, r.timing["seconds"])
    print(tag, round(n / best / 1e3, 1), "k frames/s", flush=True)
t(SequenceRunner(fe_a, res, NF, keyframe_every=10), "one context")
t(SequenceRunner(fe_a, res, NF, keyframe_every=10, input_ready=True), "one context + input-ready")
fe_b = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
t(SequenceRunner([fe_a, fe_b], res, NF, keyframe_every=10), "two contexts")
t(SequenceRunner(fe_a, pool, NF, keyframe_every=10, input_ready=True), "pcie one context + input-ready")
t(SequenceRunner(fe_a, pool, NF, keyframe_every=10), "pcie one context")
