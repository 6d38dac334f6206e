"""One-off hunt: device batches of odd sizes (the XCD work order deals frames to 8 groups) against the oracle, with
frame-to-frame links.  Usage: python tools/fuzz_batch.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from oracle import orb_oracle as oo  # noqa: E402
from orb_slam3_amd import _lib, synth  # noqa: E402
from orb_slam3_amd.features import ORBFeatureExtractor  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
bad = 0
for B, W, H, NF, L in [(1, 320, 240, 300, 4), (7, 320, 240, 300, 4), (8, 336, 256, 200, 3), (9, 320, 240, 500, 5), (17, 400, 304, 400, 4),
                       (41, 320, 240, 250, 4), (64, 320, 240, 150, 2)]:
    frames = synth.sequence(300 + B, min(B, 16), H, W)
    frames = np.concatenate([frames] * ((B + len(frames) - 1) // len(frames)))[:B].copy()
    for f in range(B):
        frames[f] = np.roll(frames[f], f, axis=1)            # every frame different
    fe = ORBFeatureExtractor(W, H, 1.2, L, 20, 7, max_batch=B, max_features=NF)
    cap = fe.geometry(NF).keypoint_capacity
    d_imgs = torch.from_numpy(frames).to(dev)
    d_kp = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()                                  # the fills run on torch's stream, the library on its own
    for rep in range(2):                                      # twice: the second call reuses every buffer
        assert lib.orbgpu_extract_batch_device(fe.handle, d_imgs.data_ptr(), W * H, W, B, NF, d_kp.data_ptr(), d_desc.data_ptr(), cap,
                                               d_n.data_ptr()) == 0
    npairs = max(B - 1, 1)
    to = torch.arange(1, B, dtype=torch.int32, device=dev) if B > 1 else torch.zeros(1, dtype=torch.int32, device=dev)
    frm = torch.arange(0, B - 1, dtype=torch.int32, device=dev) if B > 1 else torch.zeros(1, dtype=torch.int32, device=dev)
    d_m = torch.zeros((npairs, cap), dtype=torch.int32, device=dev)
    d_d = torch.zeros((npairs, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros(npairs, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    assert lib.orbgpu_match_frames_device(fe.handle, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, to.data_ptr(), frm.data_ptr(),
                                          npairs, 100, 50, 0.9, 1, d_m.data_ptr(), d_d.data_ptr(), d_nm.data_ptr()) == 0
    assert lib.orbgpu_synchronize(fe.handle) == 0
    n = d_n.cpu().numpy()
    kp = d_kp.cpu().numpy().view(_lib.KP_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy()
    m = d_m.cpu().numpy()
    p = oo.params(W, H, levels=L)
    ok = [oo.extract(p, frames[f], NF) for f in range(B)]
    for f in range(B):
        k, d = ok[f]
        good = n[f] == len(k) and all(np.array_equal(kp[f, :n[f]][x], k[x]) for x in ("x", "y", "level")) and np.array_equal(desc[f, :n[f]], d)
        if not good:
            print("EXTRACT MISMATCH", B, f)
            bad += 1
    for pi in range(B - 1):
        t, fr = pi + 1, pi
        mf, _ = oo.match_area(ok[t][0], ok[t][1], ok[fr][0], ok[fr][1], 100, 50, 0.9)
        mf, _ = oo.orientation_validate(ok[t][0]["angle"], ok[fr][0]["angle"], mf)
        if not np.array_equal(m[pi, :n[t]], mf):
            print("LINK MISMATCH", B, pi)
            bad += 1
    print("batch", B, "ok" if not bad else "", flush=True)
    fe.close()
print("done,", bad, "bad")
