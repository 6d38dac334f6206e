"""Oracle parity on exactly what bench.py times:
  * the headline workload -- 256 frames of 640x480 / 8 levels / 1000 keypoints through orbgpu_extract_batch_device +
    orbgpu_match_frames_device (255 links), every frame and every link;
  * a 64-frame 1920x1080 / 5000-keypoint batch (the three-kernel link path at its real capacity);
  * brute-force SNN at sizes that select k_snn_partial<false, 2> (n_to >= 16384), with planted matches, duplicate rows
    (ties) and contested 'from' claims.
Reference semantics: orb_feature_extractor.cpp:530-600, monocular_frame.cpp:173-189,
second_nearest_neighbor_matcher.hpp:40-71."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from orb_slam3_amd import synth

pytestmark = pytest.mark.gpu


def _bench_frames(seed, B, H, W, unique):
    base = synth.sequence(seed, min(B, unique), H, W)          # bench.py: a short seeded sequence tiled to the batch
    idx = np.arange(B) % len(base)
    return base, idx, np.ascontiguousarray(base[idx])


def _extract_and_link(oracle, W, H, NF, B, unique, seed):
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    base, idx, frames = _bench_frames(seed, B, H, W, unique)
    fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
    cap = fe.geometry(NF).keypoint_capacity
    dev = torch.device("cuda", 0)
    d_imgs = torch.from_numpy(frames).to(dev)
    d_kp = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    n_pairs = B - 1
    d_to = torch.arange(1, B, dtype=torch.int32, device=dev)
    d_from = torch.arange(0, B - 1, dtype=torch.int32, device=dev)
    d_m = torch.zeros((n_pairs, cap), dtype=torch.int32, device=dev)
    d_d = torch.zeros((n_pairs, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for rep in range(2):      # twice: the second pass runs on warm scratch, as every timed step of bench.py does
        rc = lib.orbgpu_extract_batch_device(fe.handle, d_imgs.data_ptr(), W * H, W, B, NF, d_kp.data_ptr(),
                                             d_desc.data_ptr(), cap, d_n.data_ptr())
        assert rc == 0, lib.orbgpu_last_error(fe.handle)
        rc = lib.orbgpu_match_frames_device(fe.handle, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap,
                                            d_to.data_ptr(), d_from.data_ptr(), n_pairs, 100, 50, 0.9, 1,
                                            d_m.data_ptr(), d_d.data_ptr(), d_nm.data_ptr())
        assert rc == 0, lib.orbgpu_last_error(fe.handle)
    assert lib.orbgpu_synchronize(fe.handle) == 0
    n = d_n.cpu().numpy()
    kp = d_kp.cpu().numpy().view(_lib.KP_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy()
    m, nm = d_m.cpu().numpy(), d_nm.cpu().numpy()
    fe.close()

    p = oracle.params(W, H)
    with ThreadPoolExecutor(8) as ex:                          # the C oracle releases the GIL
        ok = list(ex.map(lambda f: oracle.extract(p, base[f], NF), range(len(base))))
    for f in range(B):
        k, d = ok[idx[f]]
        assert n[f] == len(k), f"frame {f}"
        got = kp[f, :n[f]]
        for fld in ("x", "y", "level", "size"):
            assert np.array_equal(got[fld], k[fld]), f"frame {f} {fld}"
        assert np.array_equal(got["angle"], k["angle"].astype(np.float32)), f"frame {f} angle"
        assert np.array_equal(desc[f, :n[f]], d), f"frame {f} descriptors"
    links = {}
    for pi in range(n_pairs):
        t, fr = idx[pi + 1], idx[pi]
        if (t, fr) not in links:
            mf, _ = oracle.match_area(ok[t][0], ok[t][1], ok[fr][0], ok[fr][1], 100, 50, 0.9)
            links[(t, fr)], _ = oracle.orientation_validate(ok[t][0]["angle"], ok[fr][0]["angle"], mf)
        mf = links[(t, fr)]
        assert np.array_equal(m[pi, :n[pi + 1]], mf), f"pair {pi}"
        assert (m[pi, n[pi + 1]:] == -1).all(), f"pair {pi} tail"
        assert nm[pi] == (mf >= 0).sum()
    return n, nm


def test_headline_256_frames_vga(oracle):
    """BASELINE.json configs[1] as bench.py runs it (seed 1235, 32 unique frames tiled to 256)."""
    n, nm = _extract_and_link(oracle, 640, 480, 1000, 256, 32, 1235)
    assert n.mean() > 900 and nm.mean() > 100


def test_large_64_frames_1080p(oracle):
    """BASELINE.json configs[2], 1920x1080 / 5000 keypoints, 64 frames (8 unique): keypoint capacity above the fused link
    kernel's LDS limit, so the links run through k_link_partial / finalize / resolve."""
    n, nm = _extract_and_link(oracle, 1920, 1080, 5000, 64, 8, 1236)
    assert n.mean() > 4500 and nm.mean() > 300


def test_batch_4k_frames(oracle):
    """BASELINE.json configs[2], 3840x2160 / 5000 keypoints: a 4-frame device batch and its 3 links (grid link path, 1024-thread
    quadtree workgroups) against the oracle."""
    n, nm = _extract_and_link(oracle, 3840, 2160, 5000, 4, 4, 1237)
    assert n.mean() > 4500 and nm.mean() > 200


def test_large_capacity_link_grid_equals_plain_scan(oracle):
    """Above the fused link kernel's capacity orbgpu_match_frames_device bins the level-0 rows by the 64 x 48 search grid
    (k_link_grid); ORBGPU_TUNE=link_scan=1 keeps the plain predicated scan.  Same matches, distances and counts, on pairs that
    share frames in both roles (a frame linked to two others, and in both directions), with and without the rotation
    filter, for windows of 100 and 30 px; one pair is checked against the oracle as well."""
    import os
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    W, H, NF, B = 1920, 1080, 5000, 6
    frames = synth.sequence(77, B, H, W)
    dev = torch.device("cuda", 0)
    pairs = [(1, 0), (2, 1), (2, 0), (0, 2), (5, 4), (3, 3)]
    res = {}
    for mode in ("grid", "scan"):
        if mode == "scan":
            os.environ["ORBGPU_TUNE"] = "link_scan=1"
        try:
            fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
        finally:
            os.environ.pop("ORBGPU_TUNE", None)
        cap = fe.geometry(NF).keypoint_capacity
        d_imgs = torch.from_numpy(frames).to(dev)
        d_kp = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        d_n = torch.zeros(B, dtype=torch.int32, device=dev)
        d_to = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=dev)
        d_from = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        assert lib.orbgpu_extract_batch_device(fe.handle, d_imgs.data_ptr(), W * H, W, B, NF, d_kp.data_ptr(),
                                               d_desc.data_ptr(), cap, d_n.data_ptr()) == 0
        out = []
        for window, check in ((100, 1), (100, 0), (30, 1)):
            d_m = torch.full((len(pairs), cap), 9, dtype=torch.int32, device=dev)
            d_d = torch.full((len(pairs), cap), 9, dtype=torch.int32, device=dev)
            d_nm = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            for rep in range(2):
                rc = lib.orbgpu_match_frames_device(fe.handle, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap,
                                                    d_to.data_ptr(), d_from.data_ptr(), len(pairs), window, 50, 0.9, check,
                                                    d_m.data_ptr(), d_d.data_ptr(), d_nm.data_ptr())
                assert rc == 0, lib.orbgpu_last_error(fe.handle)
            assert lib.orbgpu_synchronize(fe.handle) == 0
            out.append((d_m.cpu().numpy(), d_d.cpu().numpy(), d_nm.cpu().numpy()))
        res[mode] = out
        if mode == "grid":
            n = d_n.cpu().numpy()
            kp = d_kp.cpu().numpy().view(_lib.KP_DTYPE).reshape(B, cap)
            desc = d_desc.cpu().numpy()
        fe.close()
    for (gm, gd, gn), (sm, sd, sn) in zip(res["grid"], res["scan"]):
        assert np.array_equal(gm, sm) and np.array_equal(gn, sn)
        assert np.array_equal(gd[gm >= 0], sd[sm >= 0])
    assert res["grid"][0][2].min() > 200                          # also the frame linked to itself
    to, fr = pairs[2]
    def ok(f):
        k = np.zeros(n[f], oracle.KP_DTYPE)
        for fld in ("x", "y", "angle", "size", "level"):
            k[fld] = kp[f, :n[f]][fld]
        return k, desc[f, :n[f]]
    (kt, dt_), (kf, df_) = ok(to), ok(fr)
    mf, _ = oracle.match_area(kt, dt_, kf, df_, 100, 50, 0.9)
    mf, _ = oracle.orientation_validate(kt["angle"], kf["angle"], mf)
    assert np.array_equal(res["grid"][0][0][2, :n[to]], mf)


@pytest.mark.parametrize("n_to,n_from,settings", [
    (20011, 700, ((50, 0.9), (100, 0.8), (100, 0.6), (50, 1.0))),      # n_to not a multiple of 512, one 'from' chunk group
    (100000, 4000, ((50, 0.9), (100, 0.8))),
    (20000, 20000, ((100, 0.6), (50, 0.9))),
])
def test_bruteforce_two_rows_per_lane(oracle, n_to, n_from, settings):
    """n_to >= 16384 selects k_snn_partial<false, 2> (two 'to' rows per lane, 512-row blocks): the kernel behind
    bench.py's 100k x 100k figure."""
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    rng = np.random.default_rng(n_to + n_from)
    frm = synth.descriptors(500 + n_from, n_from)
    to = synth.descriptors(600 + n_to, n_to, planted_from=frm, planted_frac=0.3, max_flips=60)
    # duplicate 'from' rows: best == second for whoever matches them -> ratio test fails for ratio <= 1
    dup = rng.choice(n_from - 1, 20, replace=False)
    frm[dup + 1] = frm[dup]
    # contested claims: the same 'to' descriptor at several rows, also across the two 256-row halves of a 512-row block
    # and across blocks -> only the earliest row keeps the 'from'
    src = rng.choice(n_to, 30, replace=False)
    for s in src:
        for off in (1, 256, 511, 512, 7777):
            to[(s + off) % n_to] = to[s]
    to[n_to - 1] = to[0]                                              # the last (partial) block against the first
    fe = ORBFeatureExtractor(640, 480, 1.2, 8, 20, 7, max_features=1000)
    dev = torch.device("cuda", 0)
    dt, df = torch.from_numpy(to).to(dev), torch.from_numpy(frm).to(dev)
    di = torch.zeros(n_to, dtype=torch.int32, device=dev)
    dd = torch.zeros(n_to, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for th, ratio in settings:
        rc = lib.orbgpu_match_bruteforce_device(fe.handle, dt.data_ptr(), n_to, df.data_ptr(), n_from, th, ratio,
                                                di.data_ptr(), dd.data_ptr())
        assert rc == 0, lib.orbgpu_last_error(fe.handle)
        assert lib.orbgpu_synchronize(fe.handle) == 0
        omf, omd = oracle.snn_match(to, frm, th, ratio)
        gi, gd = di.cpu().numpy(), dd.cpu().numpy().view(np.uint32)
        assert np.array_equal(gi, omf), (th, ratio)
        sel = omf >= 0
        assert np.array_equal(gd[sel], omd[sel])
        assert sel.sum() > 100
    fe.close()


def test_bench_two_ranks_on_one_gpu():
    """`bench.py --gpus 2` starts two ranks by itself; ORB_BENCH_REHEARSAL=1 puts both on GPU 0 over gloo (RCCL wants
    one GPU per rank), so the whole N > 1 flow -- frame sharding, barrier, MAX over ranks, the all-gather of the
    keyframe shards -- runs on a one-GPU box."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ORB_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--frames", "8", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-bruteforce", "--no-database", "--sequence-frames", "203"], env=env,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["world_size_observed"] == 2 and out["value"] > 0
    assert out["keyframe_allpairs"]["keyframes_total"] == 4 and out["keyframe_allpairs"]["matches_rank0"] > 0
    assert "not the headline" in out["config"]["workload"]
    # the offline sequence, block-sharded over the two ranks (102 + 101 frames, one halo frame, 21 keyframes gathered)
    seq = out["sequence"]
    assert seq["frames"] == 203 and seq["keyframes_total"] == 21 and seq["pcie_inclusive"]["frames_per_s"] > 0
    assert seq["keyframe_allpairs"]["matches_rank0"] > 0 and "all_gather" in seq["keyframe_allpairs"]["exchange"]
    assert seq["keyframe_allpairs_mfma_i8_exploration"]["identical_to_popcount_results"] is True
