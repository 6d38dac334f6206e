"""CPU tests: the C-ABI library loads and exports every symbol include/orbgpu.h declares, the host-side
interface mirror keeps the reference's names, and the frame-sharding / all-gather plumbing works on 2 gloo ranks."""
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_abi_exports_match_header():
    from orb_slam3_amd import _lib
    header = (ROOT / "include" / "orbgpu.h").read_text()
    declared = set(re.findall(r"\b(orbgpu_[a-z0-9_]+)\s*\(", header))
    declared -= {"orbgpu_hamming256"}  # static inline
    assert declared == set(_lib.EXPORTS)
    if not _lib.LIB_PATH.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "orb_slam3_amd" / "csrc"), "-j8"])
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None
    # every declaration cites the reference interface it replaces
    assert header.count("replaces:") >= 8


def test_product_never_touches_oracle():
    for p in (ROOT / "orb_slam3_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".cpp", ".h"):
            assert "oracle" not in p.read_text().replace("the oracle", "").replace("as in the oracle", ""), p


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from orb_slam3_amd._lib import OrbGpuError
    from orb_slam3_amd.features import ORBFeatureExtractor
    with pytest.raises(OrbGpuError):
        ORBFeatureExtractor(640, 480, 1.2, 8, 20, 7)


def test_inline_hamming_matches_oracle(oracle, tmp_path):
    src = tmp_path / "h.c"
    src.write_text('#include "orbgpu.h"\n#include <stdio.h>\nint main(){unsigned char a[32],b[32];for(int i=0;i<32;i++){a[i]=i*7+1;b[i]=255-i*3;}printf("%u\\n",orbgpu_hamming256(a,b));return 0;}\n')
    exe = tmp_path / "h"
    subprocess.check_call(["gcc", "-O2", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)])
    got = int(subprocess.check_output([str(exe)]).decode())
    a = np.array([(i * 7 + 1) & 255 for i in range(32)], np.uint8)
    b = np.array([(255 - i * 3) & 255 for i in range(32)], np.uint8)
    assert got == oracle.hamming256(a, b)


def test_frame_block_partition():
    from orb_slam3_amd.sharding import frame_block, keyframe_ids, allpairs_plan
    for n, w in ((20000, 8), (17, 4), (5, 8), (256, 1), (0, 2)):
        blocks = [frame_block(n, w, r) for r in range(w)]
        owned = [f for b in blocks for f in range(b.start, b.stop)]
        assert owned == list(range(n))
        for b in blocks:
            if b.n_owned and b.start > 0:
                assert b.halo_start == b.start - 1  # t-1 of the first owned frame is extracted locally
        kf = [k for b in blocks for k in keyframe_ids(b, 10)]
        assert kf == list(range(0, n, 10))
    plan = [p for r, (o, k) in enumerate([(0, 2), (2, 3)]) for p in allpairs_plan(o, k, 5)]
    assert sorted(plan) == sorted((i, j) for i in range(5) for j in range(5) if i != j)


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from orb_slam3_amd.sharding import frame_block, keyframe_ids, gather_keyframe_descriptors, allpairs_plan
dist.init_process_group("gloo", init_method="file://" + os.environ["RDZV_FILE"], rank=int(os.environ["RANK"]),
                        world_size=int(os.environ["WORLD_SIZE"]))   # file rendezvous: no port to collide on
rank, world = dist.get_rank(), dist.get_world_size()
n_frames, every, cap = 47, 5, 6
blk = frame_block(n_frames, world, rank)
kfs = keyframe_ids(blk, every)
g = torch.Generator().manual_seed(1)
all_desc = torch.randint(0, 256, (n_frames, cap, 32), dtype=torch.uint8, generator=g)
all_n = torch.randint(0, cap + 1, (n_frames,), dtype=torch.int32, generator=g)
desc, n, off = gather_keyframe_descriptors(all_desc[kfs], all_n[kfs])
want = list(range(0, n_frames, every))
assert torch.equal(desc, all_desc[want]) and torch.equal(n, all_n[want])
assert want[off:off + len(kfs)] == kfs
cnt = torch.tensor([len(allpairs_plan(off, len(kfs), len(want)))])
dist.all_reduce(cnt)
assert int(cnt) == len(want) * (len(want) - 1)
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_allgather_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=str(ROOT)))
    env = dict(os.environ, RDZV_FILE=str(tmp_path / "rendezvous"), WORLD_SIZE="2", GLOO_SOCKET_IFNAME="lo")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_interface_mirror_names():
    from orb_slam3_amd import features
    for name in ("Extract", "Type", "GetAcceptableSquareError", "ComputeInvariantDistances", "PredictScale",
                 "ComputeDistance", "GetScaleFactors", "GetHighThreshold", "GetLowThreshold", "Serialize"):
        assert hasattr(features.ORBFeatureExtractor, name)
    assert hasattr(features.OrientationValidator, "Validate")


def test_features_wire_format_round_trip():
    """Python mirror of operator<< / Features(std::istream &) (features.cpp:14-79), following the reference's own
    round-trip test (test/features/feature_tests.cpp:54-104): 100 keypoints, 30-byte descriptor rows."""
    import numpy as np
    from orb_slam3_amd.features import Features
    from orb_slam3_amd._lib import KP_DTYPE
    rng = np.random.default_rng(0)
    f = Features(100, 200)
    f.descriptors = rng.integers(0, 255, (100, 30), dtype=np.uint8)
    f.keypoints = np.zeros(100, KP_DTYPE)
    f.keypoints["x"], f.keypoints["y"] = rng.uniform(0, 150, 100), rng.uniform(0, 250, 100)
    f.keypoints["size"], f.keypoints["angle"], f.keypoints["level"] = 1, 23, 5
    f.undistorted_keypoints = rng.uniform(0, 250, (100, 2))
    f.undistorted_and_unprojected_keypoints = rng.uniform(0, 30, (100, 3))
    blob = f.Serialize()
    assert len(blob) == 16 + 16 + 100 * 30 + 100 * (4 + 8 + 8 + 16) + 100 * 16 + 100 * 24
    assert np.frombuffer(blob, "<i8", 2, 16).tolist() == [30, 100]      # features_width, features_height
    g = Features.Deserialize(blob)
    assert (g.image_width, g.image_height) == (100.0, 200.0)
    assert np.array_equal(g.descriptors, f.descriptors)
    for name in ("x", "y", "angle", "size", "level"):
        assert np.array_equal(g.keypoints[name], f.keypoints[name])
    assert np.array_equal(g.undistorted_keypoints, f.undistorted_keypoints)
    assert np.array_equal(g.undistorted_and_unprojected_keypoints, f.undistorted_and_unprojected_keypoints)
    empty = Features.Deserialize(Features(64, 48).Serialize())
    assert empty.Size() == 0 and empty.descriptors.shape == (0, 32)


def test_keyframe_database_stream_layout():
    """DBoW2Database::Serialize (frame/database/DBoW2/dbo_w2_database.cpp:39-51): u64 word count, per word u64 entry
    count then (u64 keyframe id, u64 count) pairs; round trip through the stream constructor (:16-32).  No GPU needed:
    only the query touches the device."""
    import io
    import struct
    from types import SimpleNamespace
    from orb_slam3_amd.database import DBoW2Database
    fe = SimpleNamespace(_lib=None, handle=None)
    db = DBoW2Database(fe, 6, with_counts=True)
    db.Append(7, {1: 0.25, 4: 0.75})
    db.Append(3, {4: 1.0}, {4: 2})
    db.Append(9, {0: 0.5, 5: 0.5})
    db.Erase(9)
    buf = io.BytesIO()
    db.Serialize(buf)
    want = struct.pack("<Q", 6) + struct.pack("<Q", 0) + struct.pack("<QQQ", 1, 7, 1) + struct.pack("<Q", 0) * 2 + \
        struct.pack("<QQQQQ", 2, 3, 2, 7, 1) + struct.pack("<Q", 0)
    assert buf.getvalue() == want
    buf.seek(0)
    back = DBoW2Database.Deserialize(buf, fe, {7: {1: 0.25, 4: 0.75}, 3: {4: 1.0}})
    assert back.inverted_file == db.inverted_file
    import pytest
    with pytest.raises(ValueError):
        DBoW2Database.Deserialize(io.BytesIO(want[:-4]), fe, {})
    with pytest.raises(ValueError):
        db.Append(1, {6: 1.0})
    with pytest.raises(ValueError):
        DBoW2Database(fe, 6).Append(1, {2: 1.0}, {2: 3})       # counts need with_counts=True


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (plain child processes, one per
    GPU), which rendezvous, reduce their times with MAX and let rank 0 print the line; --dry-run keeps it to that
    control flow (gloo, no GPU, no compute) so that it runs here."""
    import json
    import subprocess
    import sys
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size_observed"] == 2 and out["dry_run"] is True
    # a failing rank fails the launcher: without a GPU every rank of the real bench stops with "needs a GPU"
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--frames", "8"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "rank exit codes" in r.stderr


def test_kernel_reciprocal_divisions_are_exact():
    """k_fast_cells (csrc/k_fast.hip, quick_round) divides by a 16-bit reciprocal: with M = floor(65536 / d) + 1 or + 2
    (v_rcp_f32's estimate of 65536 / d is within 0.02 of it) it takes n * M >> 16 for n / d.  Exhaustive over the
    ranges the kernel uses: thread id / quads per row (d <= 36, n < 256), 256 / quads per row, and
    (rows - 1) / rows per pass (d <= 256, n <= 63)."""
    for d in range(1, 37):
        for M in (65536 // d + 1, 65536 // d + 2):
            for n in range(0, 257):
                assert (n * M) >> 16 == n // d, (n, d, M)
    for d in range(1, 257):
        for M in (65536 // d + 1, 65536 // d + 2):
            for n in range(0, 64):
                assert (n * M) >> 16 == n // d, (n, d, M)


def test_packed_score_network_identity():
    """k_fast_cells scores a pixel from the ring pixels themselves (fast_score16_full): with d = v - ring,
    max over arcs of min(d) = v - min over arcs of max(ring) and max over arcs of min(-d) = max over arcs of min(ring) - v.
    Checked against the definition on random rings."""
    rng = np.random.default_rng(5)
    for _ in range(2000):
        ring = rng.integers(0, 256, 16)
        v = int(rng.integers(0, 256))
        d = v - ring
        arcs = [[(k + i) % 16 for i in range(9)] for k in range(16)]
        m_dark = max(min(d[a]) for a in arcs)
        m_bright = max(min(-d[a]) for a in arcs)
        a_min = min(max(ring[a]) for a in arcs)
        b_max = max(min(ring[a]) for a in arcs)
        assert m_dark == v - a_min and m_bright == b_max - v
