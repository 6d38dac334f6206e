#!/usr/bin/env python3
"""Writes tests/golden/*.npz from the CPU oracle (oracle/) on small seeded inputs.

The reference holds no golden vectors for this path and cannot run here, so these fixtures pin the
ORACLE (regressions show up in tests/test_golden.py) and give the GPU box committed data to compare against.
Rerun after an intentional oracle change:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import orb_oracle as oo  # noqa: E402
from orb_slam3_amd import synth  # noqa: E402

HERE = Path(__file__).resolve().parent
CASES = [("seq_200x160_l4", 200, 160, 4, 300, 41), ("seq_320x240_l8", 320, 240, 8, 500, 42)]


def main():
    for name, w, h, levels, nfeat, seed in CASES:
        frames = synth.sequence(seed, 2, h, w)
        p = oo.params(w, h, levels=levels)
        k0, d0, ex0 = oo.extract(p, frames[0], nfeat, dump=True)
        k1, d1 = oo.extract(p, frames[1], nfeat)
        mf, md = oo.match_area(k1, d1, k0, d0, 100, 50, 0.9)
        mfo, nd = oo.orientation_validate(k1["angle"], k0["angle"], mf)
        bf, bd = oo.snn_match(d1, d0, 100, 0.8)
        np.savez_compressed(HERE / f"{name}.npz", frames=frames, params=np.array([w, h, levels, nfeat]),
                            kp0=k0, desc0=d0, kp1=k1, desc1=d1, pyramid0=ex0["pyramid"], score0=ex0["score"],
                            n_cands0=np.array(ex0["n_cands"]), link=mf, link_dist=md, link_filtered=mfo,
                            n_discarded=np.array(nd), brute=bf, brute_dist=bd)
        print(name, len(k0), len(k1), int((mf >= 0).sum()), int((mfo >= 0).sum()), int((bf >= 0).sum()))


if __name__ == "__main__":
    main()
