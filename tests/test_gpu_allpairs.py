"""GPU parity of orbgpu_match_allpairs_device (all-pairs keyframe matching) against the CPU oracle:
brute-force SNN with ratio / threshold / uniqueness per (row keyframe, column keyframe) pair, optional rotation
histogram.  Descriptor sets are synthetic with planted near-duplicates and engineered ties."""
import numpy as np
import pytest

from orb_slam3_amd import synth

pytestmark = pytest.mark.gpu


def _sets(seed, n_kf, cap, n_min, n_max):
    rng = np.random.default_rng(seed)
    n = rng.integers(n_min, n_max + 1, n_kf).astype(np.int32)
    desc = np.zeros((n_kf, cap, 32), np.uint8)
    kp = np.zeros((n_kf, cap), dtype=None)
    base = rng.integers(0, 256, (n_max, 32), dtype=np.uint8)   # a shared "map": keyframes see noisy copies of it
    from orb_slam3_amd._lib import KP_DTYPE
    kp = np.zeros((n_kf, cap), KP_DTYPE)
    for k in range(n_kf):
        pick = rng.permutation(n_max)[:n[k]]
        d = base[pick].copy()
        flips = rng.integers(0, 40, n[k])
        for i in range(n[k]):
            bits = rng.choice(256, flips[i], replace=False)
            for b in bits:
                d[i, b >> 3] ^= 1 << (b & 7)
        # engineered ties: duplicate a few rows so best == second and equal-distance claims occur
        if n[k] > 8:
            d[1] = d[0]
            d[n[k] - 1] = d[n[k] - 2]
        desc[k, :n[k]] = d
        kp[k, :n[k]]["angle"] = rng.choice([10.0, 12.5, 40.0, 200.0, 359.5], n[k]).astype(np.float32) + (k % 3) * 30
    return kp, desc, n


@pytest.mark.parametrize("engine", ["popcount", "mfma_i8"])
@pytest.mark.parametrize("cap,n_min,n_max,n_kf", [(256, 0, 200, 5), (1024, 900, 1024, 3), (1536, 1100, 1536, 2)])
def test_allpairs_matches_oracle(oracle, cap, n_min, n_max, n_kf, engine):
    """both engines (orbgpu_set_bruteforce_engine): the popcount kernel -- the default and the contract path -- and the
    exploratory i8 matrix-core kernel (k_allpairs_mfma), which takes a second and third turn over the 'from' rows for
    keyframes of more than 512 / 1024 keypoints"""
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    fe = ORBFeatureExtractor(160, 120, 1.2, 3, 20, 7, max_features=100)
    assert lib.orbgpu_set_bruteforce_engine(fe.handle, _lib.ENGINE_MFMA_I8 if engine == "mfma_i8" else _lib.ENGINE_POPCOUNT) == 0
    kp, desc, n = _sets(5 + cap, n_kf, cap, n_min, n_max)
    if n_min == 0:
        n[1] = 0   # an empty keyframe
    dev = torch.device("cuda", 0)
    d_kp = torch.from_numpy(kp.view(np.uint8).reshape(n_kf, cap, 32)).to(dev)
    d_desc = torch.from_numpy(desc).to(dev)
    d_n = torch.from_numpy(n).to(dev)
    for rows, row_offset, check_orientation, want_lists in (((0, n_kf), 0, 0, True), ((1, n_kf), 1, 1, True),
                                                            ((0, n_kf), -1, 1, False)):
        r0, r1 = rows
        nr = r1 - r0
        d_nm = torch.full((nr, n_kf), -7, dtype=torch.int32, device=dev)
        d_m = torch.full((nr, n_kf, cap), -7, dtype=torch.int32, device=dev) if want_lists else None
        torch.cuda.synchronize()     # the fills run on torch's stream, the library on its own
        rc = lib.orbgpu_match_allpairs_device(fe.handle, d_kp[r0:].data_ptr(), d_desc[r0:].data_ptr(), d_n[r0:].data_ptr(), nr,
                                              d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), n_kf, cap, row_offset,
                                              50, 0.9, check_orientation, d_nm.data_ptr(),
                                              d_m.data_ptr() if want_lists else None)
        assert rc == 0, lib.orbgpu_last_error(fe.handle)
        assert lib.orbgpu_synchronize(fe.handle) == 0
        nm = d_nm.cpu().numpy()
        m = d_m.cpu().numpy() if want_lists else None
        for r in range(nr):
            for c in range(n_kf):
                gr = r0 + r
                if (row_offset >= 0 and row_offset + r == c) or n[gr] == 0 or n[c] == 0:
                    assert nm[r, c] == 0
                    if want_lists:
                        assert (m[r, c] == -1).all()
                    continue
                mf, _ = oracle.snn_match(desc[gr, :n[gr]], desc[c, :n[c]], 50, 0.9)
                if check_orientation:
                    mf, _ = oracle.orientation_validate(kp[gr, :n[gr]]["angle"], kp[c, :n[c]]["angle"], mf)
                assert nm[r, c] == (mf >= 0).sum(), (r, c)
                if want_lists:
                    assert np.array_equal(m[r, c, :n[gr]], mf), (r, c)
                    assert (m[r, c, n[gr]:] == -1).all()
    fe.close()


def test_allpairs_errors():
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    fe = ORBFeatureExtractor(160, 120, 1.2, 3, 20, 7, max_features=100)
    one = 4096  # any non-null address: the call must be rejected before it is used
    assert lib.orbgpu_match_allpairs_device(fe.handle, None, one, one, 2, None, one, one, 2, 8192, -1, 50, 0.9, 0, one,
                                            None) == _lib.E_CAPACITY
    assert lib.orbgpu_match_allpairs_device(fe.handle, None, one, one, 2, None, one, one, 2, 512, -1, 50, 0.9, 1, one,
                                            None) == _lib.E_INVALID     # orientation without keypoints
    assert lib.orbgpu_match_allpairs_device(fe.handle, None, None, one, 2, None, one, one, 2, 512, -1, 50, 0.9, 0, one,
                                            None) == _lib.E_INVALID
    assert lib.orbgpu_match_allpairs_device(fe.handle, None, None, None, 0, None, None, None, 0, 512, -1, 50, 0.9, 0,
                                            None, None) == 0
    fe.close()


_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from orb_slam3_amd import synth, _lib
from orb_slam3_amd.features import ORBFeatureExtractor
from orb_slam3_amd.sharding import frame_block, keyframe_ids, match_keyframes_allpairs
# two ranks share the one GPU of the test box; RCCL needs one GPU per rank.  File rendezvous: no fixed port to collide on
dist.init_process_group("gloo", init_method="file://" + os.environ["RDZV_FILE"], rank=int(os.environ["RANK"]),
                        world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
W, H, NF, n_frames, every = 320, 240, 300, 14, 2
dev = torch.device("cuda", 0)
lib = _lib.load()
fe = ORBFeatureExtractor(W, H, 1.2, 6, 20, 7, max_batch=n_frames, max_features=NF)
cap = fe.geometry(NF).keypoint_capacity

def extract(ids):
    imgs = np.stack([synth.frame(900 + i, H, W) for i in ids])
    d_imgs = torch.from_numpy(imgs).to(dev)
    kp = torch.zeros((len(ids), cap, 32), dtype=torch.uint8, device=dev)
    desc = torch.zeros((len(ids), cap, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros(len(ids), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()         # the fills run on torch's stream, the library on its own
    _lib.check(lib.orbgpu_extract_batch_device(fe.handle, d_imgs.data_ptr(), W * H, W, len(ids), NF, kp.data_ptr(),
                                              desc.data_ptr(), cap, n.data_ptr()), fe.handle)
    _lib.check(lib.orbgpu_synchronize(fe.handle), fe.handle)
    return kp, desc, n

blk = frame_block(n_frames, world, rank)
kfs = keyframe_ids(blk, every)
kp, desc, n = extract(kfs)
nm, m, off = match_keyframes_allpairs(fe, kp, desc, n, 50, 0.9, True, want_lists=True)
# single-process answer for the same rows
all_kfs = list(range(0, n_frames, every))
akp, adesc, an = extract(all_kfs)
assert all_kfs[off:off + len(kfs)] == kfs
ref_nm = torch.zeros((len(all_kfs), len(all_kfs)), dtype=torch.int32, device=dev)
ref_m = torch.zeros((len(all_kfs), len(all_kfs), cap), dtype=torch.int32, device=dev)
# the zero fills above run on torch's stream, the library on its own: without ordering, a fill can land after the
# library's writes (it did, once in 25 runs).  An event instead of a device-wide synchronisation:
ev = torch.cuda.Event()
ev.record()
_lib.check(lib.orbgpu_wait_event(fe.handle, ev.cuda_event), fe.handle)
_lib.check(lib.orbgpu_match_allpairs_device(fe.handle, akp.data_ptr(), adesc.data_ptr(), an.data_ptr(), len(all_kfs),
                                           akp.data_ptr(), adesc.data_ptr(), an.data_ptr(), len(all_kfs), cap, 0, 50, 0.9,
                                           1, ref_nm.data_ptr(), ref_m.data_ptr()), fe.handle)
_lib.check(lib.orbgpu_synchronize(fe.handle), fe.handle)
assert torch.equal(nm, ref_nm[off:off + len(kfs)])
assert torch.equal(m, ref_m[off:off + len(kfs)])
assert int(nm.sum()) > 0
tot = nm.sum().cpu().to(torch.int64)
dist.all_reduce(tot)
assert int(tot) == int(ref_nm.sum())
dist.destroy_process_group()
print("rank", rank, "ok", int(tot))
"""


def test_allpairs_two_ranks_one_gpu(tmp_path):
    """Frame-sharded extraction + ONE all-gather + row-sharded all-pairs matching on two ranks equals the
    single-process result (gloo rendezvous; both ranks use the test box's single GPU)."""
    import os, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    script = tmp_path / "rank_worker.py"
    script.write_text(_RANK_WORKER.format(root=str(root)))
    env = dict(os.environ, RDZV_FILE=str(tmp_path / "rendezvous"), WORLD_SIZE="2", GLOO_SOCKET_IFNAME="lo")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=110)[0].decode() for p in procs]
    if not all(p.returncode == 0 for p in procs):
        pytest.fail("\n".join(f"--- rank {r} (exit {p.returncode}) ---\n{o}" for r, (p, o) in enumerate(zip(procs, outs))))


def test_gather_over_rccl_single_rank(tmp_path):
    """The exchange step on the real backend: a one-rank "nccl" (= RCCL) group, initialised the way bench.py does it,
    (file rendezvous here)
    runs the all_gather / all_gather_into_tensor calls of gather_keyframe_rows on device tensors of the three dtypes
    the path sends (uint8 records and descriptors, int32 counts).  More ranks need more GPUs than the test box has;
    the multi-rank control flow is covered by the gloo tests."""
    import subprocess
    import sys
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, os.getcwd())\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', init_method='file://' + sys.argv[1], rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "from orb_slam3_amd.sharding import gather_keyframe_rows\n"
        "g = torch.Generator().manual_seed(3)\n"
        "desc = torch.randint(0, 256, (5, 40, 32), dtype=torch.uint8, generator=g).cuda()\n"
        "n = torch.randint(0, 41, (5,), dtype=torch.int32, generator=g).cuda()\n"
        "(d2, n2), off = gather_keyframe_rows([desc, n], _force_collective=True)\n"
        "assert off == 0 and torch.equal(d2, desc) and torch.equal(n2, n) and d2.is_cuda\n"
        "t = torch.tensor([1.5], dtype=torch.float64, device='cuda'); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()\n"
        "assert t.item() == 1.5\n"
        "dist.destroy_process_group()\n"
        "print('rccl ok')\n")
    root = str(__import__("pathlib").Path(__file__).resolve().parents[1])
    out = subprocess.run([sys.executable, str(script), str(tmp_path / "rendezvous")], capture_output=True, text=True, timeout=100, cwd=root)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stderr[-2000:]
