import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from oracle import orb_oracle as oo
from orb_slam3_amd import synth
from orb_slam3_amd._lib import OrbGpuError
from orb_slam3_amd.features import Features, ORBFeatureExtractor
bad = 0
for (w, h, scale, levels, nf) in [(96, 64, 1.2, 2, 50), (100, 80, 1.2, 3, 80), (120, 100, 1.3, 3, 100), (75, 75, 1.2, 1, 30), (131, 77, 1.15, 4, 60),
                                   (257, 65, 1.2, 2, 100), (64, 200, 1.2, 3, 50), (203, 199, 1.7, 3, 200), (89, 151, 1.25, 3, 100), (72, 72, 1.1, 3, 40)]:
    img = synth.frame(w * 7 + h, h, w)
    try:
        p = oo.params(w, h, scale, levels, 20, 7)
        okp, odesc = oo.extract_literal(p, img, nf)
    except Exception as e:
        print((w, h, scale, levels), "oracle refuses:", str(e)[:80]); okp = None
    try:
        fe = ORBFeatureExtractor(w, h, scale, levels, 20, 7, max_features=nf)
    except OrbGpuError as e:
        print((w, h, scale, levels), "library refuses:", str(e)[:100]); continue
    ft = Features(w, h)
    rc = fe.Extract(img, ft, nf)
    if okp is None:
        print((w, h, scale, levels), "library accepted, rc", rc); fe.close(); continue
    same = len(ft.keypoints) == len(okp) and np.array_equal(ft.descriptors, odesc) and all(
        (a.X() == b[0] and a.Y() == b[1]) if hasattr(a, 'X') else True for a, b in zip(ft.keypoints, okp))
    print((w, h, scale, levels), "n =", len(okp), "same" if same else "DIFFER")
    bad += 0 if same else 1
    fe.close()
print("bad", bad)
