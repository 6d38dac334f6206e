"""One-off hunt: seeded random SNN matching problems (brute force and windowed, with duplicated descriptors so that
distance ties and best == second cases occur) against the oracle.  Usage: python tools/fuzz_match.py FIRST_SEED COUNT."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401,E402  (its HIP runtime first)
from oracle import orb_oracle as oo  # noqa: E402
from orb_slam3_amd import synth  # noqa: E402
from orb_slam3_amd.features import ORBFeatureExtractor, SNNMatcher  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
fe = ORBFeatureExtractor(320, 240, 1.2, 4, 20, 7, max_features=300)
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n_from = int(rng.choice([0, 1, 2, 63, 64, 65, 300, 1000, 2500, 4097]))
    n_to = int(rng.choice([0, 1, 5, 64, 257, 900, 3000]))
    frm = synth.descriptors(seed, n_from)
    if n_from > 4:                                        # exact duplicates inside 'from': ties between candidates
        dup = rng.integers(0, n_from, n_from // 6)
        frm[rng.integers(0, n_from, len(dup))] = frm[dup]
    to = synth.descriptors(seed + 7, n_to, planted_from=frm if n_from else None, planted_frac=0.6, max_flips=int(rng.choice([0, 3, 40, 90])))
    th = int(rng.choice([0, 30, 50, 100, 256, 400]))
    ratio = float(rng.choice([0.6, 0.7, 0.9, 1.0]))
    kw = {}
    if rng.random() < 0.7:
        kw["to_xy"] = rng.uniform(0, 640, (n_to, 2)).round(int(rng.integers(0, 3)))
        kw["from_xy"] = rng.uniform(0, 640, (n_from, 2)).round(int(rng.integers(0, 3)))
        kw["to_window"] = rng.integers(0, 300, n_to).astype(np.uint64)
        kw["from_level"] = rng.integers(0, 8, n_from).astype(np.int32)
        lo = rng.integers(0, 8, n_to)
        kw["to_min_level"] = lo.astype(np.int32)
        kw["to_max_level"] = (lo + rng.integers(-1, 3, n_to)).astype(np.int32)
        if rng.random() < 0.5:
            kw["to_active"] = (rng.random(n_to) < 0.8).astype(np.uint8)
    om, od = oo.snn_match(to, frm, th, ratio, **kw)
    gm, gd = SNNMatcher(ratio, th, fe).MatchRaw(to, frm, **kw)
    if not (np.array_equal(om, gm) and np.array_equal(od[om >= 0], gd[gm >= 0])):
        print("MISMATCH", seed, n_to, n_from, th, ratio, sorted(kw), int((om != gm).sum()))
        bad += 1
    if seed % 50 == 0:
        print("..", seed, int((om >= 0).sum()), flush=True)
print("done", count, "cases,", bad, "bad")
