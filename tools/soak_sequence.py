"""Soak: the sequence runner with two contexts alternating (and with the input-ready hint) many times over; every run must
reproduce the one-context run bit for bit (a missing event between the two contexts' streams would show up as a rare
difference).  Usage: python tools/soak_sequence.py [REPS]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401
from orb_slam3_amd.features import ORBFeatureExtractor  # noqa: E402
from orb_slam3_amd.sequence import PooledSource, ResidentSource, SequenceRunner  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H, NF, B = 640, 480, 1000, 64
fe_a = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
fe_b = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
pool = PooledSource(17, pool_frames=48, height=H, width=W)
res = ResidentSource(pool, torch.device("cuda", 0), repeat=4)
n = 9 * B + 13
one = SequenceRunner(fe_a, pool, NF).run(n, keep_links=True)
bad = 0
for rep in range(reps):
    for src in (pool, res):
        for fes, hint in (([fe_a, fe_b], False), ([fe_a, fe_b], True), (fe_a, True)):
            got = SequenceRunner(fes, src, NF, input_ready=hint).run(n, keep_links=True)
            ok = np.array_equal(one.n_keypoints, got.n_keypoints) and np.array_equal(one.links, got.links) and \
                bool((one.kf_n == got.kf_n).all())
            if ok:
                kn = one.kf_n.cpu().numpy()
                a, b = one.kf_desc.cpu().numpy(), got.kf_desc.cpu().numpy()
                ok = all(np.array_equal(a[q, :kn[q]], b[q, :kn[q]]) for q in range(len(kn)))
            if not ok:
                bad += 1
                print("DIFFERS", rep, "resident" if src is res else "pooled", "two" if isinstance(fes, list) else "one", hint, flush=True)
    if rep % 5 == 0:
        print("..", rep, flush=True)
print("done:", reps, "repetitions,", bad, "differ")
