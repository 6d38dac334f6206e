import sys, time
sys.path.insert(0, "/root/repo")
import torch
from orb_slam3_amd.features import ORBFeatureExtractor
from orb_slam3_amd.sequence import PooledSource, SequenceRunner
fe = ORBFeatureExtractor(640, 480, 1.2, 8, 20, 7, max_batch=256, max_features=1000)
pool = PooledSource(1240, pool_frames=64)
for nc in (1, 2, 3, 4):
    for ring in (3, 4):
        r = SequenceRunner(fe, pool, 1000, ring=ring, copy_streams=nc)
        r.run(768)
        best = min(r.run(20000).timing["seconds"] for _ in range(3))
        print("copy streams", nc, "ring", ring, "frames/s", round(20000 / best), "GB/s", round(20000 * 307200 / best / 1e9, 1))
