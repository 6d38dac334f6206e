"""One-off hunt: many seeded random extractor configurations (tests/test_gpu_extract.py::_fuzz_case) against the
oracle.  Usage: python tools/fuzz_extract.py FIRST_SEED COUNT.  Prints the failing configurations."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402  (its HIP runtime first)
from oracle import orb_oracle as oo  # noqa: E402
from orb_slam3_amd import synth  # noqa: E402
from orb_slam3_amd._lib import OrbGpuError  # noqa: E402
from orb_slam3_amd.features import Features, ORBFeatureExtractor  # noqa: E402
from test_gpu_extract import _fuzz_case  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    w, h, nfeat, scale, levels, fi, fm, kind = cfg = _fuzz_case(seed)
    img = synth.frame(seed, h, w)
    if kind == "low":
        img = (img.astype(np.float32) * 0.12 + 100).astype(np.uint8)
    elif kind == "noise":
        img = np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)
    p = oo.params(w, h, scale, levels, fi, fm)
    try:
        okp, odesc = oo.extract(p, img, nfeat)
    except ValueError:
        try:
            ORBFeatureExtractor(w, h, scale, levels, fi, fm, max_features=nfeat)
            print("NOT REFUSED", seed, cfg)
            bad += 1
        except OrbGpuError:
            pass
        continue
    fe = ORBFeatureExtractor(w, h, scale, levels, fi, fm, max_features=nfeat)
    ft = Features(w, h)
    rc = fe.Extract(img, ft, nfeat)
    ok = rc == 0 and ft.Size() == len(okp) and all(np.array_equal(ft.keypoints[f], okp[f]) for f in ("x", "y", "level", "size")) \
        and np.array_equal(ft.keypoints["angle"], okp["angle"].astype(np.float32)) and np.array_equal(ft.descriptors, odesc)
    if not ok:
        print("MISMATCH", seed, cfg, rc, ft.Size(), len(okp))
        bad += 1
    fe.close()
    if seed % 50 == 0:
        print("..", seed, flush=True)
print("done", count, "cases,", bad, "bad")
