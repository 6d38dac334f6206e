"""The offline sequence path (BASELINE.json configs[4]; reference loop: RunDataset main.cpp:334-411, link
monocular_frame.cpp:173-200) through orb_slam3_amd.sequence.SequenceRunner: streamed batches with the upload ring, links
across batch and rank boundaries, keyframes kept on the device, all-pairs keyframe matching -- against the oracle and
against the single-call paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, NF, B, EVERY = 640, 480, 1000, 256, 10


@pytest.fixture(scope="module")
def runner():
    from orb_slam3_amd.features import ORBFeatureExtractor
    from orb_slam3_amd.sequence import ProceduralSource, SequenceRunner
    fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
    src = ProceduralSource(4242, H, W)
    r = SequenceRunner(fe, src, NF, keyframe_every=EVERY)
    yield r
    fe.close()


@pytest.fixture(scope="module")
def full_run(runner):
    return runner.run(600, keep_links=True, keep_features=True)


def _oracle_frame(oracle, runner, cache, i):
    if i not in cache:
        cache[i] = oracle.extract(oracle.params(W, H), runner.src.seq.frame(i), NF)
    return cache[i]


def test_sequence_600_frames_against_oracle(oracle, runner, full_run):
    """600 frames in batches of 256 (last batch partial: 88): every 37th frame, and both frames of every link that
    crosses a batch boundary, extracted by the oracle from the same procedural frame; those links and the links of the
    sampled frames against the oracle's window matcher + rotation filter."""
    res = full_run
    assert res.timing["batches"] == 3 and res.timing["frames_extracted"] == 600
    assert res.n_keypoints.shape == (600,) and res.n_links[0] == 0
    cache = {}
    boundary = [256, 512]
    frames = sorted(set(range(0, 600, 37)) | {b - 1 for b in boundary} | set(boundary) | {599})
    for i in frames:
        k, d = _oracle_frame(oracle, runner, cache, i)
        n = int(res.n_keypoints[i])
        assert n == len(k), f"frame {i}"
        got = res.keypoints[i, :n]
        for fld in ("x", "y", "level", "size"):
            assert np.array_equal(got[fld], k[fld]), f"frame {i} {fld}"
        assert np.array_equal(got["angle"], k["angle"].astype(np.float32)), f"frame {i} angle"
        assert np.array_equal(res.descriptors[i, :n], d), f"frame {i} descriptors"
    link_frames = sorted(set(boundary) | {37, 74, 300, 599})
    for t in link_frames:
        kt, dt = _oracle_frame(oracle, runner, cache, t)
        kf, df = _oracle_frame(oracle, runner, cache, t - 1)
        mf, _ = oracle.match_area(kt, dt, kf, df, 100, 50, 0.9)
        mf, _ = oracle.orientation_validate(kt["angle"], kf["angle"], mf)
        assert np.array_equal(res.links[t, :len(kt)], mf), f"link {t} -> {t - 1}"
        assert (res.links[t, len(kt):] == -1).all()
        assert res.n_links[t] == (mf >= 0).sum()
        assert res.n_links[t] > 100                            # consecutive frames really do match
    assert (res.links[0] == -1).all()
    assert res.keyframe_ids == list(range(0, 600, EVERY))
    kfn = res.kf_n.cpu().numpy()
    assert np.array_equal(kfn, res.n_keypoints[::EVERY])
    kfd = res.kf_desc.cpu().numpy()
    for q in (0, 25, 26, 51, 59):                             # keyframes 250 / 260 sit either side of a batch boundary
        assert np.array_equal(kfd[q, :kfn[q]], res.descriptors[q * EVERY, :kfn[q]])


def test_sequence_equals_one_batch_calls(runner, full_run):
    """The streamed run equals plain orbgpu_extract_batch_device + orbgpu_match_frames_device calls on resident frames
    (what tests/test_gpu_headline.py pins against the oracle frame by frame), over a stretch that spans a batch
    boundary: frames 200 .. 320."""
    import torch
    from orb_slam3_amd import _lib
    lib, fe = runner.lib, runner.fe
    a, b = 200, 321
    n = b - a
    frames = np.zeros((n, H, W), np.uint8)
    runner.src.fill(a, n, frames)
    dev = runner.dev
    cap = runner.cap
    d_img = torch.from_numpy(frames).to(dev)
    d_kp = torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(n, dtype=torch.int32, device=dev)
    d_to = torch.arange(1, n, dtype=torch.int32, device=dev)
    d_from = torch.arange(0, n - 1, dtype=torch.int32, device=dev)
    d_m = torch.zeros((n - 1, cap), dtype=torch.int32, device=dev)
    d_d = torch.zeros((n - 1, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros(n - 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    assert lib.orbgpu_extract_batch_device(fe.handle, d_img.data_ptr(), W * H, W, n, NF, d_kp.data_ptr(), d_desc.data_ptr(),
                                           cap, d_n.data_ptr()) == 0
    assert lib.orbgpu_match_frames_device(fe.handle, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, d_to.data_ptr(),
                                          d_from.data_ptr(), n - 1, 100, 50, 0.9, 1, d_m.data_ptr(), d_d.data_ptr(),
                                          d_nm.data_ptr()) == 0
    assert lib.orbgpu_synchronize(fe.handle) == 0
    res = full_run
    cnt = d_n.cpu().numpy()
    assert np.array_equal(cnt, res.n_keypoints[a:b])
    kp, desc = d_kp.cpu().numpy().view(_lib.KP_DTYPE).reshape(n, cap), d_desc.cpu().numpy()
    for f in range(n):                     # rows beyond a frame's count are not written by the extractor
        same_kp = np.array_equal(kp[f, :cnt[f]], res.keypoints[a + f, :cnt[f]])
        same_desc = np.array_equal(desc[f, :cnt[f]], res.descriptors[a + f, :cnt[f]])
        assert same_kp and same_desc, f"frame {a + f}"
    same_links = np.array_equal(d_m.cpu().numpy(), res.links[a + 1:b])
    assert same_links
    assert np.array_equal(d_nm.cpu().numpy(), res.n_links[a + 1:b])


def test_sequence_sharded_blocks_equal_single_run(runner, full_run):
    """The same 600 frames as the blocks of 7 ranks (ragged: 86 / 86 / 86 / 86 / 86 / 85 / 85): every rank's owned frames,
    links (the first one through its halo frame) and keyframes equal the single run's."""
    res = full_run
    world = 7
    kf_seen = []
    for rank in range(world):
        part = runner.run(600, rank=rank, world=world, keep_links=True, keep_features=True)
        blk = part.block
        assert part.timing["frames_extracted"] == blk.n_owned + (1 if rank > 0 else 0)
        s, e = blk.start, blk.stop
        assert np.array_equal(part.n_keypoints, res.n_keypoints[s:e])
        for f in range(s, e):
            c = res.n_keypoints[f]
            same_kp = np.array_equal(part.keypoints[f - s, :c], res.keypoints[f, :c])
            same_desc = np.array_equal(part.descriptors[f - s, :c], res.descriptors[f, :c])
            assert same_kp and same_desc, f"rank {rank} frame {f}"
        assert np.array_equal(part.n_links, res.n_links[s:e])          # includes the link through the halo frame
        same_links = np.array_equal(part.links, res.links[s:e])
        assert same_links, f"rank {rank}"
        kf_seen += part.keyframe_ids
        q0 = res.keyframe_ids.index(part.keyframe_ids[0])
        pn, rn = part.kf_n.cpu().numpy(), res.kf_n.cpu().numpy()[q0:q0 + len(part.keyframe_ids)]
        assert np.array_equal(pn, rn)
        pd, rd = part.kf_desc.cpu().numpy(), res.kf_desc.cpu().numpy()[q0:q0 + len(part.keyframe_ids)]
        assert all(np.array_equal(pd[q, :pn[q]], rd[q, :pn[q]]) for q in range(len(pn))), f"rank {rank} keyframes"
    assert kf_seen == res.keyframe_ids


def test_sequence_keyframe_allpairs(oracle, runner, full_run):
    """All-pairs matching of the run's 60 keyframes: the runner's row-blocked path = one match_keyframes_allpairs call
    = the oracle on a few pairs."""
    import torch
    from orb_slam3_amd.sharding import match_keyframes_allpairs
    res = full_run
    nm_blocks, off = runner.match_keyframes(res, row_block=16)
    nm_one, lists, off1 = match_keyframes_allpairs(runner.fe, res.kf_kp, res.kf_desc, res.kf_n, 50, 0.9, True, True)
    assert off == 0 and off1 == 0
    a, b = nm_blocks.cpu().numpy(), nm_one.cpu().numpy()
    assert a.shape == (60, 60) and np.array_equal(a, b)
    assert (np.diag(a) == 0).all() and a.sum() > 0
    lists = lists.cpu().numpy()
    kfn = res.kf_n.cpu().numpy()
    for (r, c) in ((0, 1), (1, 0), (25, 26), (59, 58), (3, 40)):
        kr, dr = res.keypoints[r * EVERY, :kfn[r]], res.descriptors[r * EVERY, :kfn[r]]
        kc, dc = res.keypoints[c * EVERY, :kfn[c]], res.descriptors[c * EVERY, :kfn[c]]
        mf, _ = oracle.snn_match(dr, dc, 50, 0.9)
        mf, _ = oracle.orientation_validate(kr["angle"].astype(np.float64), kc["angle"].astype(np.float64), mf)
        assert np.array_equal(lists[r, c, :kfn[r]], mf), (r, c)
        assert a[r, c] == (mf >= 0).sum()


def test_sequence_pooled_and_resident_sources(runner):
    """The throughput sources of bench.py: frames handed over as slices of a pinned pool (uploaded without a staging
    copy) and of a device-resident pool; both give what the staged procedural source gives for the same frames, with a
    pool shorter than a batch (wrapping runs) and a sequence that ends in a partial batch."""
    from orb_slam3_amd.sequence import PooledSource, ResidentSource, SequenceRunner
    fe = runner.fe
    pool = PooledSource(77, pool_frames=24, height=H, width=W)
    out = []
    for src in (pool, ResidentSource(pool, runner.dev), ResidentSource(pool, runner.dev, repeat=11)):  # the last: batches extracted in place
        r = SequenceRunner(fe, src, NF, keyframe_every=EVERY)
        out.append(r.run(300, keep_links=True, keep_features=True))
    a, b, c_ = out
    assert np.array_equal(a.n_keypoints, c_.n_keypoints) and np.array_equal(a.n_links, c_.n_links) and np.array_equal(a.links, c_.links)
    assert c_.timing["bytes_uploaded"] < 300 * W * H  # 24 x 11 = 264 frames: the first batch of 256 is one slice
    assert np.array_equal(a.n_keypoints, b.n_keypoints) and np.array_equal(a.n_links, b.n_links)
    for f in range(300):
        c = a.n_keypoints[f]
        same = np.array_equal(a.descriptors[f, :c], b.descriptors[f, :c]) and np.array_equal(a.links[f], b.links[f])
        assert same, f"frame {f}"
    # frame i = pool frame i % 24: same keypoints whenever the pool index repeats; the link of frame 24 goes back to 23
    c = a.n_keypoints[3]
    assert a.n_keypoints[27] == c and a.n_keypoints[3 + 24 * 11] == c
    same = np.array_equal(a.descriptors[3, :c], a.descriptors[27, :c]) and np.array_equal(a.descriptors[3, :c], a.descriptors[3 + 24 * 11, :c])
    assert same
    same = np.array_equal(a.links[5], a.links[29])
    assert same and a.n_links[5] > 100
    assert a.timing["bytes_uploaded"] == 300 * W * H


def test_two_contexts_alternating(runner):
    """batches alternating between two contexts of the library (the throughput arrangement of bench.py: the tail of a
    batch beside the head of the next): per-frame counts, links across every batch boundary (the carry crosses the two
    output sets), features and keyframes equal the one-context run's, with a partial last batch"""
    from orb_slam3_amd.features import ORBFeatureExtractor
    from orb_slam3_amd.sequence import PooledSource, SequenceRunner
    pool = PooledSource(91, pool_frames=40, height=H, width=W)
    fe_b = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
    n = 3 * B + 77
    one = SequenceRunner(runner.fe, pool, NF, keyframe_every=EVERY).run(n, keep_links=True, keep_features=True)
    two_runner = SequenceRunner([runner.fe, fe_b], pool, NF, keyframe_every=EVERY)
    for rep in range(2):  # the second run starts on sets that hold the first run's last batches
        two = two_runner.run(n, keep_links=True, keep_features=True)
        assert np.array_equal(one.n_keypoints, two.n_keypoints) and np.array_equal(one.n_links, two.n_links)
        assert np.array_equal(one.links, two.links)
        for f in range(n):  # rows beyond a frame's count hold whatever the output set held before
            c = one.n_keypoints[f]
            same = np.array_equal(one.descriptors[f, :c], two.descriptors[f, :c]) and np.array_equal(one.keypoints[f, :c], two.keypoints[f, :c])
            assert same, f"frame {f}"
        assert one.keyframe_ids == two.keyframe_ids
        assert bool((one.kf_n == two.kf_n).all())
        kn = one.kf_n.cpu().numpy()
        a_d, b_d, a_k, b_k = one.kf_desc.cpu().numpy(), two.kf_desc.cpu().numpy(), one.kf_kp.cpu().numpy(), two.kf_kp.cpu().numpy()
        for q in range(len(kn)):
            same = np.array_equal(a_d[q, :kn[q]], b_d[q, :kn[q]]) and np.array_equal(a_k[q, :kn[q]], b_k[q, :kn[q]])
            assert same, f"keyframe {q}"
    for b in (B, 2 * B, 3 * B):  # the links into the first frame of a batch come from the other context's outputs
        assert one.n_links[b] > 50
    # the input-ready hint (a call's head beside the previous call's tail), one context and two
    for fes in (runner.fe, [runner.fe, fe_b]):
        got = SequenceRunner(fes, pool, NF, keyframe_every=EVERY, input_ready=True).run(n, keep_links=True)
        assert np.array_equal(one.n_keypoints, got.n_keypoints) and np.array_equal(one.n_links, got.n_links)
        assert np.array_equal(one.links, got.links)
        assert bool((one.kf_n == got.kf_n).all())
    # 2 ranks, two contexts each: the block + halo arithmetic is untouched
    parts = [two_runner.run(n, rank=r, world=2, keep_links=True) for r in range(2)]
    assert np.array_equal(np.concatenate([p.n_links for p in parts]), one.n_links)
    assert np.array_equal(np.concatenate([p.links for p in parts]), one.links)
    fe_b.close()
