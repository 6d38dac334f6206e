"""One-off hunt: random windowed-search problems (sizes, image shapes, radii, clustered / degenerate positions) through
orbgpu_match_window_grid_segments_device against orbgpu_match_window_segments_device, and the first segment against the
oracle.  Usage: python tools/fuzz_grid.py FIRST_SEED COUNT"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from oracle import orb_oracle as oo  # noqa: E402
from orb_slam3_amd import _lib  # noqa: E402
from orb_slam3_amd.features import ORBFeatureExtractor  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
lib = _lib.load()
fe = ORBFeatureExtractor(640, 480, 1.2, 8, 20, 7, max_features=1000)
dev = torch.device("cuda", 0)
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    W, H = float(rng.integers(80, 4000)), float(rng.integers(80, 2200))
    S = int(rng.integers(1, 6))
    cap = int(rng.choice([1, 7, 64, 65, 300, 1000, 2500]))
    sizes = [(int(rng.integers(0, cap + 1)), int(rng.integers(0, cap + 1))) for _ in range(S)]
    kind = rng.choice(["uniform", "cluster", "one_cell", "border", "line"])
    N = S * cap

    def pts(n):
        if kind == "uniform":
            return rng.uniform((-30, -30), (W + 30, H + 30), (n, 2))
        if kind == "cluster":
            c = rng.uniform((0, 0), (W, H), (max(1, n // 50), 2))
            return c[rng.integers(0, len(c), n)] + rng.normal(0, 12, (n, 2))
        if kind == "one_cell":
            return rng.uniform((W / 2, H / 2), (W / 2 + W / 64, H / 2 + H / 48), (n, 2))
        if kind == "border":
            p = rng.uniform((0, 0), (W, H), (n, 2))
            p[:, int(rng.integers(0, 2))] = rng.choice([0.0, -5.0, W, H, W + 3], n)
            return p
        p = rng.uniform((0, 0), (W, H), (n, 2))
        p[:, 1] = np.floor(p[:, 1] / (H / 48)) * (H / 48)       # on cell boundaries
        return p

    from_xy, to_xy = pts(N), pts(N)
    from_desc = rng.integers(0, 256, (N, 32), dtype=np.uint8)
    to_desc = rng.integers(0, 256, (N, 32), dtype=np.uint8)
    k = N // 2
    sel = rng.choice(N, k, replace=False)
    to_desc[sel] = from_desc[sel]
    to_desc[sel, :3] ^= rng.integers(0, 256, (k, 3)).astype(np.uint8)
    to_xy[sel] = from_xy[sel] + rng.uniform(-6, 6, (k, 2))
    if N > 20:
        from_desc[sel[1]] = from_desc[sel[0]]; from_xy[sel[1]] = from_xy[sel[0]]
        to_desc[sel[3]] = to_desc[sel[2]]; to_xy[sel[3]] = to_xy[sel[2]]
    to_lv = rng.integers(0, 8, N).astype(np.int32)
    from_lv = rng.integers(0, 8, N).astype(np.int32)
    radius = float(rng.choice([0, 1, 5, 15, 30, 100, 400, 5000]))
    win = (np.float64(radius) * np.float64(1.2) ** to_lv).astype(np.uint64)
    mode = int(rng.integers(0, 3))
    if mode == 0:
        lmin, lmax = np.maximum(to_lv - 1, 0).astype(np.int32), (to_lv + 1).astype(np.int32)
    elif mode == 1:
        lmin, lmax = np.full(N, -1, np.int32), np.full(N, -1, np.int32)
    else:
        lmin, lmax = np.zeros(N, np.int32), np.zeros(N, np.int32)
    active = (rng.uniform(size=N) < 0.8).astype(np.uint8)
    segs = np.zeros(S, _lib.SEGMENT_DTYPE)
    for s, (nt, nf) in enumerate(sizes):
        segs[s] = (s * cap, nt, s * cap, nf)
    t = {k_: torch.from_numpy(v).to(dev) for k_, v in dict(
        to_desc=to_desc, from_desc=from_desc, to_xy=to_xy, from_xy=from_xy, win=win.view(np.int64), lmin=lmin, lmax=lmax,
        from_lv=from_lv, active=active, segs=segs.view(np.int32).reshape(S, 4)).items()}
    o = [torch.full((N,), 5, dtype=torch.int32, device=dev) for _ in range(4)]
    th, ratio = int(rng.choice([50, 100, 256])), float(rng.choice([0.6, 0.9, 1.0]))
    a = _lib.WindowArgs(t["to_desc"].data_ptr(), t["to_xy"].data_ptr(), t["win"].data_ptr(), t["lmin"].data_ptr(),
                        t["lmax"].data_ptr(), t["active"].data_ptr(), N, t["from_desc"].data_ptr(), t["from_xy"].data_ptr(),
                        t["from_lv"].data_ptr(), N, th, ratio)
    max_to, max_from = max(x for x, _ in sizes), max(y for _, y in sizes)
    torch.cuda.synchronize()
    assert lib.orbgpu_match_window_segments_device(fe.handle, C.byref(a), t["segs"].data_ptr(), S, max_to, o[0].data_ptr(),
                                                   o[1].data_ptr()) == 0
    assert lib.orbgpu_match_window_grid_segments_device(fe.handle, C.byref(a), t["segs"].data_ptr(), S, max_to, max_from,
                                                        W, H, o[2].data_ptr(), o[3].data_ptr()) == 0
    rc = lib.orbgpu_synchronize(fe.handle)
    ok = rc == 0 and torch.equal(o[0], o[2]) and torch.equal(o[1], o[3])
    nt, nf = sizes[0]
    if ok and nt:
        omf, _ = oo.snn_match(to_desc[:nt], from_desc[:nf], th, ratio, to_xy=to_xy[:nt], to_window=win[:nt],
                              to_min_level=lmin[:nt], to_max_level=lmax[:nt], to_active=active[:nt], from_xy=from_xy[:nf],
                              from_level=from_lv[:nf])
        ok = np.array_equal(o[2].cpu().numpy()[:nt], omf)
    if not ok:
        bad += 1
        print("BAD", seed, W, H, S, cap, sizes, kind, radius, mode, th, ratio, rc)
    if (seed - first) % 50 == 49:
        print("..", seed + 1, bad, flush=True)
print(f"done {count} cases, {bad} bad")
