"""GPU tests of the device-resident batch entry points used by bench.py: orbgpu_extract_batch_device and
orbgpu_match_frames_device, checked per frame / per pair against the CPU oracle."""
import numpy as np
import pytest

from orb_slam3_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,NF,B", [(320, 240, 400, 6), (640, 480, 3000, 3)])
def test_batch_device_extract_and_link(oracle, W, H, NF, B):
    """The second case has a keypoint capacity above what the fused per-pair link kernel keeps in LDS, so it takes
    the three-kernel path through global scratch."""
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    frames = synth.sequence(77, B, H, W)
    fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
    cap = fe.geometry(NF).keypoint_capacity
    dev = torch.device("cuda", 0)
    d_imgs = torch.from_numpy(frames).to(dev)
    d_kp = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()         # the fills run on torch's stream, the library on its own
    rc = lib.orbgpu_extract_batch_device(fe.handle, d_imgs.data_ptr(), W * H, W, B, NF, d_kp.data_ptr(),
                                         d_desc.data_ptr(), cap, d_n.data_ptr())
    assert rc == 0, lib.orbgpu_last_error(fe.handle)
    to = torch.tensor([(i + 1) % B for i in range(B)], dtype=torch.int32, device=dev)
    frm = torch.tensor(list(range(B)), dtype=torch.int32, device=dev)
    npairs = B
    d_m = torch.zeros((npairs, cap), dtype=torch.int32, device=dev)
    d_d = torch.zeros((npairs, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros(npairs, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for check_orientation in (0, 1):
        rc = lib.orbgpu_match_frames_device(fe.handle, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap,
                                            to.data_ptr(), frm.data_ptr(), npairs, 100, 50, 0.9, check_orientation,
                                            d_m.data_ptr(), d_d.data_ptr(), d_nm.data_ptr())
        assert rc == 0, lib.orbgpu_last_error(fe.handle)
        assert lib.orbgpu_synchronize(fe.handle) == 0
        n = d_n.cpu().numpy()
        kp = d_kp.cpu().numpy().view(_lib.KP_DTYPE).reshape(B, cap)
        desc = d_desc.cpu().numpy()
        m = d_m.cpu().numpy()
        nm = d_nm.cpu().numpy()
        p = oracle.params(W, H)
        ok = [oracle.extract(p, frames[f], NF) for f in range(B)]
        for f in range(B):
            k, d = ok[f]
            assert n[f] == len(k)
            assert np.array_equal(kp[f, :n[f]]["x"], k["x"]) and np.array_equal(kp[f, :n[f]]["y"], k["y"])
            assert np.array_equal(kp[f, :n[f]]["level"], k["level"])
            assert np.array_equal(kp[f, :n[f]]["angle"], k["angle"].astype(np.float32))
            assert np.array_equal(desc[f, :n[f]], d)
        for pi, (t, fr) in enumerate(zip(to.cpu().numpy(), frm.cpu().numpy())):
            mf, _ = oracle.match_area(ok[t][0], ok[t][1], ok[fr][0], ok[fr][1], 100, 50, 0.9)
            if check_orientation:
                mf, _ = oracle.orientation_validate(ok[t][0]["angle"], ok[fr][0]["angle"], mf)
            assert np.array_equal(m[pi, :n[t]], mf), f"pair {pi}"
            assert nm[pi] == (mf >= 0).sum()
    fe.close()


def test_link_keypoints_in_any_order(oracle):
    """orbgpu_match_frames_device does not rely on the extractor's by-level packing: with the keypoints of both frames
    shuffled (levels interleaved, so no block of the large-capacity path is skipped) it still gives the oracle's links."""
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    W, H, NF, B = 640, 480, 3000, 2
    frames = synth.sequence(78, B, H, W)
    fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
    cap = fe.geometry(NF).keypoint_capacity
    p = oracle.params(W, H)
    ok = [oracle.extract(p, frames[f], NF) for f in range(B)]
    rng = np.random.default_rng(5)
    kp = np.zeros((B, cap), _lib.KP_DTYPE)
    desc = np.zeros((B, cap, 32), np.uint8)
    n = np.zeros(B, np.int32)
    shuffled = []
    for f in range(B):
        k, d = ok[f]
        perm = rng.permutation(len(k))
        k, d = k[perm], d[perm]
        shuffled.append((k, d))
        n[f] = len(k)
        for name in ("x", "y", "level", "size"):
            kp[f, :n[f]][name] = k[name]
        kp[f, :n[f]]["angle"] = k["angle"].astype(np.float32)
        desc[f, :n[f]] = d
    dev = torch.device("cuda", 0)
    d_kp = torch.from_numpy(kp.view(np.uint8).reshape(B, cap, -1)).to(dev)
    d_desc = torch.from_numpy(desc).to(dev)
    d_n = torch.from_numpy(n).to(dev)
    to = torch.tensor([1, 0], dtype=torch.int32, device=dev)
    frm = torch.tensor([0, 1], dtype=torch.int32, device=dev)
    d_m = torch.zeros((2, cap), dtype=torch.int32, device=dev)
    d_d = torch.zeros((2, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros(2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for check_orientation in (0, 1):
        rc = lib.orbgpu_match_frames_device(fe.handle, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap,
                                            to.data_ptr(), frm.data_ptr(), 2, 100, 50, 0.9, check_orientation,
                                            d_m.data_ptr(), d_d.data_ptr(), d_nm.data_ptr())
        assert rc == 0, lib.orbgpu_last_error(fe.handle)
        assert lib.orbgpu_synchronize(fe.handle) == 0
        m, nm = d_m.cpu().numpy(), d_nm.cpu().numpy()
        for pi, (t, fr) in enumerate(((1, 0), (0, 1))):
            mf, _ = oracle.match_area(shuffled[t][0], shuffled[t][1], shuffled[fr][0], shuffled[fr][1], 100, 50, 0.9)
            if check_orientation:
                mf, _ = oracle.orientation_validate(shuffled[t][0]["angle"], shuffled[fr][0]["angle"], mf)
            assert np.array_equal(m[pi, :n[t]], mf), f"pair {pi}"
            assert nm[pi] == (mf >= 0).sum() and nm[pi] > 0
    fe.close()


@pytest.mark.parametrize("W,H,row_stride,base_off,frame_pad,in_place", [
    (324, 244, 324, 0, 0, False),      # width not a multiple of 16, tight rows: level 0 is copied into the slab
    (324, 244, 336, 0, 64, True),      # rows padded to the slab pitch (pad bytes hold junk), gap between frames: read in place
    (320, 240, 320, 16, 48, True),     # tight rows are already the slab layout; base 16 bytes into the allocation
    (320, 240, 320, 4, 0, False),      # base not 16-byte aligned: copied
    (320, 240, 352, 0, 0, False),      # row stride wider than the pitch: copied
])
def test_batch_device_layouts(oracle, W, H, row_stride, base_off, frame_pad, in_place):
    """orbgpu_extract_batch_device reads level 0 where the caller put it when the layout allows (zero copy) and copies
    it otherwise; both give the oracle's result, and the debug dump returns the caller's level 0 either way."""
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    B, NF = 3, 300
    frames = synth.sequence(91, B, H, W)
    fe = ORBFeatureExtractor(W, H, 1.2, 4, 20, 7, max_batch=B, max_features=NF)
    cap = fe.geometry(NF).keypoint_capacity
    dev = torch.device("cuda", 0)
    frame_stride = row_stride * H + frame_pad
    host = np.full(base_off + B * frame_stride, 255, np.uint8)       # junk everywhere the images are not
    for f in range(B):
        v = host[base_off + f * frame_stride: base_off + f * frame_stride + row_stride * H].reshape(H, row_stride)
        v[:, :W] = frames[f]
    d_imgs = torch.from_numpy(host).to(dev)
    d_kp = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    assert (d_imgs.data_ptr() % 256) == 0
    torch.cuda.synchronize()         # the fills run on torch's stream, the library on its own
    rc = lib.orbgpu_extract_batch_device(fe.handle, d_imgs.data_ptr() + base_off, frame_stride, row_stride, B, NF,
                                         d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    assert rc == 0, lib.orbgpu_last_error(fe.handle)
    assert lib.orbgpu_synchronize(fe.handle) == 0
    assert np.array_equal(d_imgs.cpu().numpy(), host)                # the input is never written
    n = d_n.cpu().numpy()
    kp = d_kp.cpu().numpy().view(_lib.KP_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy()
    p = oracle.params(W, H, levels=4)
    for f in range(B):
        k, d, ex = oracle.extract(p, frames[f], NF, dump=True)
        assert n[f] == len(k) and len(k) > 50
        for fld in ("x", "y", "level", "size"):
            assert np.array_equal(kp[f, :n[f]][fld], k[fld]), fld
        assert np.array_equal(kp[f, :n[f]]["angle"], k["angle"].astype(np.float32))
        assert np.array_equal(desc[f, :n[f]], d)
        dump = fe.debug_dump(f)
        assert np.array_equal(dump["pyramid"][:W * H].reshape(H, W), frames[f])
        assert np.array_equal(dump["pyramid"], ex["pyramid"]) and np.array_equal(dump["blurred"], ex["blurred"])
    fe.close()


@pytest.mark.parametrize("W,H,pad", [(320, 240, 0), (324, 244, 0), (320, 240, 32)])
def test_host_batch_threaded_path(oracle, W, H, pad):
    """orbgpu_extract_batch with enough frames for its multi-threaded staging / bulk download path (>= 8): tight
    rows that equal the slab pitch, a width that needs re-pitching, and padded rows; one frame is featureless."""
    from orb_slam3_amd.features import ORBFeatureExtractor
    B, NF = 13, 250
    frames = synth.sequence(93, B, H, W)
    frames[5][:] = 77                                   # no corners: an empty slot in the middle of the batch
    imgs = []
    for f in range(B):
        buf = np.full((H, W + pad), 200, np.uint8)
        buf[:, :W] = frames[f]
        imgs.append(buf[:, :W])                         # row stride W + pad
    fe = ORBFeatureExtractor(W, H, 1.2, 4, 20, 7, max_batch=B, max_features=NF)
    if pad == 0:
        got = fe.ExtractBatch(imgs, NF)
    else:                                               # ExtractBatch makes rows contiguous; pass the padded rows as they are
        import ctypes as C
        from orb_slam3_amd import _lib
        from orb_slam3_amd.features import Features
        cap = fe.geometry(NF).keypoint_capacity
        kp = np.zeros((B, cap), _lib.KP_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        ptrs = (C.c_void_p * B)(*[im.ctypes.data for im in imgs])
        rc = fe._lib.orbgpu_extract_batch(fe.handle, ptrs, B, W + pad, NF, _lib.ptr(kp), _lib.ptr(desc), cap, _lib.ptr(n))
        assert rc == 0, fe._lib.orbgpu_last_error(fe.handle)
        got = []
        for f in range(B):
            ft = Features(W, H)
            ft.keypoints, ft.descriptors = kp[f, :n[f]], desc[f, :n[f]]
            got.append(ft)
    p = oracle.params(W, H, levels=4)
    for f in range(B):
        k, d = oracle.extract(p, frames[f], NF)
        assert got[f].Size() == len(k) and (len(k) > 50) == (f != 5)
        for fld in ("x", "y", "level", "size"):
            assert np.array_equal(got[f].keypoints[fld], k[fld]), fld
        assert np.array_equal(got[f].keypoints["angle"], k["angle"].astype(np.float32))
        assert np.array_equal(got[f].descriptors, d)
    fe.close()


def test_error_behaviour():
    from orb_slam3_amd._lib import OrbGpuError
    from orb_slam3_amd.features import ORBFeatureExtractor, Features
    fe = ORBFeatureExtractor(160, 120, 1.2, 3, 20, 7, max_features=100)
    ft = Features(160, 120)
    assert fe.Extract(np.zeros((0, 0), np.uint8), ft, 100) == -1          # empty image -> -1 like the reference
    assert fe.Extract(np.full((120, 160), 90, np.uint8), ft, 100) == 0 and ft.Size() == 0  # flat image: no corners
    with pytest.raises(OrbGpuError):
        fe.Extract(np.zeros((100, 100), np.uint8), ft, 100)                # wrong size
    with pytest.raises(OrbGpuError):
        fe.Extract(np.zeros((120, 160), np.uint8), ft, 5000)               # above max_features
    fe.close()
    with pytest.raises(OrbGpuError):
        ORBFeatureExtractor(64, 64, 1.2, 8, 20, 7)                          # top levels smaller than a FAST cell
    ORBFeatureExtractor(640, 480, 1.2, 8, 20, 7, max_features=9000).close()


def test_concurrent_callers(oracle):
    """The reference calls the extractor from its main thread and the matchers from three threads at once
    (SURVEY 8b): a context serialises its own calls, separate contexts run side by side.  Four threads, two of them
    sharing one context, each extracting and matching repeatedly; every result must equal the single-threaded one."""
    import threading
    from orb_slam3_amd.features import Features, ORBFeatureExtractor, SNNMatcher
    W, H, NF = 320, 240, 300
    frames = synth.sequence(95, 4, H, W)
    p = oracle.params(W, H, levels=4)
    want = [oracle.extract(p, fr, NF) for fr in frames]
    shared = ORBFeatureExtractor(W, H, 1.2, 4, 20, 7, max_features=NF)
    own = [ORBFeatureExtractor(W, H, 1.2, 4, 20, 7, max_features=NF) for _ in range(2)]
    errors = []

    def worker(fe, tid):
        try:
            for it in range(6):
                i = (tid + it) % 4
                ft = Features(W, H)
                assert fe.Extract(frames[i], ft, NF) == 0
                k, d = want[i]
                assert ft.Size() == len(k) and np.array_equal(ft.descriptors, d)
                assert np.array_equal(ft.keypoints["x"], k["x"]) and np.array_equal(ft.keypoints["y"], k["y"])
                j = (i + 1) % 4
                m, _ = SNNMatcher(0.9, 50, fe).MatchRaw(d, want[j][1])
                om, _ = oracle.snn_match(d, want[j][1], 50, 0.9)
                assert np.array_equal(m, om)
        except Exception as e:   # noqa: BLE001 - reported by the main thread
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(fe, t)) for t, fe in enumerate([shared, shared, own[0], own[1]])]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for fe in [shared] + own:
        fe.close()
    assert not errors, errors


def test_caller_stream_and_capacity_flag(oracle):
    """orbgpu_set_stream: with the context on torch's stream, fills, extraction, linking and torch's reads need no
    synchronisation at all.  Then the asynchronous capacity error: a frame with more keypoints than `capacity` is not
    written, orbgpu_synchronize reports ORBGPU_E_CAPACITY, and the link matcher treats that frame as empty."""
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    W, H, NF, B = 320, 240, 400, 4
    frames = synth.sequence(79, B, H, W)
    frames[2][:, :] = frames[2][:, :] // 8 * 8                  # frame 2: blocky, fewer corners than the others
    frames[2][::2] = 90
    fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
    cap = fe.geometry(NF).keypoint_capacity
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        assert lib.orbgpu_set_stream(fe.handle, st.cuda_stream) == 0
        for rep in range(3):                                        # repeated: a race would need luck to hide three times
            d_imgs = torch.from_numpy(frames).to(dev, non_blocking=True)
            d_kp = torch.full((B, cap, 32), 255, dtype=torch.uint8, device=dev)
            d_desc = torch.full((B, cap, 32), 255, dtype=torch.uint8, device=dev)
            d_n = torch.full((B,), -7, dtype=torch.int32, device=dev)
            to = torch.tensor([1, 2, 3], dtype=torch.int32, device=dev)
            frm = torch.tensor([0, 1, 2], dtype=torch.int32, device=dev)
            d_m = torch.full((3, cap), 99, dtype=torch.int32, device=dev)
            d_d = torch.full((3, cap), 99, dtype=torch.int32, device=dev)
            d_nm = torch.full((3,), 99, dtype=torch.int32, device=dev)
            rc = lib.orbgpu_extract_batch_device(fe.handle, d_imgs.data_ptr(), W * H, W, B, NF, d_kp.data_ptr(),
                                                 d_desc.data_ptr(), cap, d_n.data_ptr())
            assert rc == 0, lib.orbgpu_last_error(fe.handle)
            rc = lib.orbgpu_match_frames_device(fe.handle, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap,
                                                to.data_ptr(), frm.data_ptr(), 3, 100, 50, 0.9, 1, d_m.data_ptr(),
                                                d_d.data_ptr(), d_nm.data_ptr())
            assert rc == 0, lib.orbgpu_last_error(fe.handle)
            n = d_n.cpu().numpy()                                   # torch's own stream-ordered copies
            desc = d_desc.cpu().numpy()
            m, nm = d_m.cpu().numpy(), d_nm.cpu().numpy()
            p = oracle.params(W, H)
            ok = [oracle.extract(p, frames[f], NF) for f in range(B)]
            for f in range(B):
                assert n[f] == len(ok[f][0]) and np.array_equal(desc[f, :n[f]], ok[f][1])
            for pi in range(3):
                mf, _ = oracle.match_area(ok[pi + 1][0], ok[pi + 1][1], ok[pi][0], ok[pi][1], 100, 50, 0.9)
                mf, _ = oracle.orientation_validate(ok[pi + 1][0]["angle"], ok[pi][0]["angle"], mf)
                assert np.array_equal(m[pi, :n[pi + 1]], mf) and nm[pi] == (mf >= 0).sum()
        assert lib.orbgpu_synchronize(fe.handle) == 0
        assert lib.orbgpu_set_stream(fe.handle, None) == 0          # back on the context's own stream
    torch.cuda.synchronize()
    # ---- capacity smaller than some frames need ----------------------------------------------------------
    small = int(np.sort(n)[1])                                      # at least two frames do not fit, at least one does
    assert n.min() <= small < n.max()
    d_kp = torch.full((B, small, 32), 255, dtype=torch.uint8, device=dev)
    d_desc = torch.full((B, small, 32), 255, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_m = torch.full((3, small), 99, dtype=torch.int32, device=dev)
    d_d = torch.full((3, small), 99, dtype=torch.int32, device=dev)
    d_nm = torch.full((3,), 99, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    rc = lib.orbgpu_extract_batch_device(fe.handle, d_imgs.data_ptr(), W * H, W, B, NF, d_kp.data_ptr(),
                                         d_desc.data_ptr(), small, d_n.data_ptr())
    assert rc == 0                                                  # asynchronous: the violation is found on the device
    rc = lib.orbgpu_match_frames_device(fe.handle, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), small,
                                        to.data_ptr(), frm.data_ptr(), 3, 100, 50, 0.9, 1, d_m.data_ptr(), d_d.data_ptr(),
                                        d_nm.data_ptr())
    assert rc == 0
    assert lib.orbgpu_synchronize(fe.handle) == _lib.E_CAPACITY
    assert b"capacity" in lib.orbgpu_last_error(fe.handle)
    assert lib.orbgpu_synchronize(fe.handle) == 0                  # reported once, then cleared
    n2 = d_n.cpu().numpy()
    assert np.array_equal(n2, n)                                    # the counts say what each frame needs
    desc2, m2, nm2 = d_desc.cpu().numpy(), d_m.cpu().numpy(), d_nm.cpu().numpy()
    for f in range(B):
        if n[f] <= small:
            assert np.array_equal(desc2[f, :n[f]], ok[f][1])
        else:
            assert (desc2[f] == 255).all()                          # not written
    for pi in range(3):
        if n[pi] > small or n[pi + 1] > small:
            assert nm2[pi] == 0 and (m2[pi] == -1).all()            # an overflowed frame counts as empty
    fe.close()


@pytest.mark.parametrize("switch", ["early_l0=0", "early_levels=2", "early_levels=3", "no_overlap=1", "resize=tiles", "slices=2", "qt_presort_minpx=0",
                                    "qt_presort_minpx=0;slices=2", "qt_presort_minpx=0;ps_blur=early",
                                    "resize_nest=4:3", "patch_blur=1"])
def test_schedules_give_identical_results(monkeypatch, switch):
    """the stage schedule (level 0's FAST launch beside the resize chain, blur beside quadtree and orientation, frame
    slices), the resize kernels and the quadtree presort are choices of speed only: each alternative writes the same
    bytes as the default"""
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    W, H, NF, B = 417, 311, 600, 11
    frames = synth.sequence(91, B, H, W)
    dev = torch.device("cuda", 0)
    d_imgs = torch.from_numpy(frames).to(dev)

    def run(env):
        if env:
            monkeypatch.setenv("ORBGPU_TUNE", env)     # one variable, "key=value;key=value"
        fe = ORBFeatureExtractor(W, H, 1.2, 8, 20, 7, max_batch=B, max_features=NF)
        cap = fe.geometry(NF).keypoint_capacity
        d_kp = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        d_n = torch.zeros(B, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for _ in range(2):  # twice: the second call follows the first one's kernels on every stream
            rc = lib.orbgpu_extract_batch_device(fe.handle, d_imgs.data_ptr(), W * H, W, B, NF, d_kp.data_ptr(),
                                                 d_desc.data_ptr(), cap, d_n.data_ptr())
            assert rc == 0, lib.orbgpu_last_error(fe.handle)
        assert lib.orbgpu_synchronize(fe.handle) == 0
        out = d_kp.cpu().numpy(), d_desc.cpu().numpy(), d_n.cpu().numpy()
        fe.close()
        monkeypatch.delenv("ORBGPU_TUNE", raising=False)
        return out

    ref = run(None)
    alt = run(switch)
    assert ref[2].min() > 50
    for a, b in zip(ref, alt):
        assert np.array_equal(a, b)


def test_input_ready_gives_identical_results():
    """orbgpu_set_input_ready: the head of a call (pyramid + FAST) runs beside the previous call's descriptor and link
    kernels.  Two different batches alternating through one context (a head that started too early, or a blur that
    overwrote the slab the previous descriptors still read, would mix them up), with the link call in between, first with
    the images complete at call time, then with an input event recorded behind a copy on another stream: every call's
    outputs equal the plain-order ones"""
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    w, h, nf, b = 640, 480, 600, 24
    seq = synth.sequence(555, 2 * b, h, w)
    dev = torch.device("cuda", 0)
    imgs = [torch.from_numpy(np.ascontiguousarray(seq[i * b:(i + 1) * b])).to(dev) for i in range(2)]
    fe = ORBFeatureExtractor(w, h, 1.2, 8, 20, 7, max_batch=b, max_features=nf)
    cap = fe.geometry(nf).keypoint_capacity
    hd = fe.handle
    to = torch.arange(1, b, dtype=torch.int32, device=dev)
    frm = torch.arange(0, b - 1, dtype=torch.int32, device=dev)

    def outputs(n):
        return [(torch.zeros((b, cap, 32), dtype=torch.uint8, device=dev), torch.zeros((b, cap, 32), dtype=torch.uint8, device=dev),
                 torch.zeros(b, dtype=torch.int32, device=dev), torch.full((b - 1, cap), -7, dtype=torch.int32, device=dev),
                 torch.zeros((b - 1, cap), dtype=torch.int32, device=dev), torch.zeros(b - 1, dtype=torch.int32, device=dev))
                for _ in range(n)]

    def call(img, o):
        kp, desc, n, mi, md, mc = o
        _lib.check(lib.orbgpu_extract_batch_device(hd, img.data_ptr(), w * h, w, b, nf, kp.data_ptr(), desc.data_ptr(), cap, n.data_ptr()), hd)
        _lib.check(lib.orbgpu_match_frames_device(hd, kp.data_ptr(), desc.data_ptr(), n.data_ptr(), cap, to.data_ptr(), frm.data_ptr(),
                                                  b - 1, 100, 50, 0.9, 1, mi.data_ptr(), md.data_ptr(), mc.data_ptr()), hd)

    n_calls = 6
    ref = outputs(n_calls)
    torch.cuda.synchronize()
    for i in range(n_calls):
        call(imgs[i & 1], ref[i])
    _lib.check(lib.orbgpu_synchronize(hd), hd)

    def same(a, b_):
        n = a[2].cpu().numpy()
        if not (torch.equal(a[2], b_[2]) and torch.equal(a[5], b_[5]) and torch.equal(a[3], b_[3])):
            return False
        ad, bd, ak, bk = a[1].cpu().numpy(), b_[1].cpu().numpy(), a[0].cpu().numpy(), b_[0].cpu().numpy()
        return all(np.array_equal(ad[f, :n[f]], bd[f, :n[f]]) and np.array_equal(ak[f, :n[f]], bk[f, :n[f]]) for f in range(b))

    # (1) images complete when the call is made
    got = outputs(n_calls)
    torch.cuda.synchronize()
    _lib.check(lib.orbgpu_set_input_ready(hd, 1, None), hd)
    for i in range(n_calls):
        call(imgs[i & 1], got[i])
    _lib.check(lib.orbgpu_synchronize(hd), hd)
    for i in range(n_calls):
        assert same(ref[i], got[i]), f"call {i}"
    # (2) images written by a copy on another stream, the call made at once with the copy's event
    got = outputs(n_calls)
    ring = [torch.zeros_like(imgs[0]) for _ in range(2)]
    side = torch.cuda.Stream(dev)
    evs = [torch.cuda.Event() for _ in range(n_calls)]
    free = [torch.cuda.Event() for _ in range(n_calls)]
    torch.cuda.synchronize()
    for i in range(n_calls):
        with torch.cuda.stream(side):
            if i >= 2:
                side.wait_event(free[i - 2])   # the call that read this ring slot has finished
            ring[i & 1].copy_(imgs[i & 1], non_blocking=True)
            evs[i].record(side)
        _lib.check(lib.orbgpu_set_input_ready(hd, 1, evs[i].cuda_event), hd)
        call(ring[i & 1], got[i])
        free[i].record(torch.cuda.ExternalStream(int(lib.orbgpu_stream(hd)), device=dev))
    _lib.check(lib.orbgpu_synchronize(hd), hd)
    _lib.check(lib.orbgpu_set_input_ready(hd, 0, None), hd)
    for i in range(n_calls):
        assert same(ref[i], got[i]), f"call {i} (event)"
    # plain order again on the same context
    got = outputs(2)
    for i in range(2):
        call(imgs[i & 1], got[i])
    _lib.check(lib.orbgpu_synchronize(hd), hd)
    assert same(ref[0], got[0]) and same(ref[1], got[1])
    fe.close()


@pytest.mark.gpu
def test_input_ready_working_sets_mixed_calls():
    """Under orbgpu_set_input_ready a context alternates between two working sets (pyramid, cells, candidates).  THREE
    different batches rotate through it, so that the two calls that share a set never hold the same images (a tail that
    ran before its head had finished would otherwise reproduce the right answer from two calls back), and the calls are
    mixed: the hint switched off and on again in between, a synchronisation, an orbgpu_trim (second set freed and
    allocated again), a call with stage timing (never piped) and a call with another keypoint budget.  Every call's
    outputs equal those of the same call on a context that never saw the hint."""
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import ORBFeatureExtractor
    lib = _lib.load()
    w, h, nf, b = 640, 480, 500, 64   # (batches long enough for a call's head to really run beside the previous tail)
    seq = synth.sequence(909, 3 * b, h, w)
    dev = torch.device("cuda", 0)
    imgs = [torch.from_numpy(np.ascontiguousarray(seq[i * b:(i + 1) * b])).to(dev) for i in range(3)]
    fe = ORBFeatureExtractor(w, h, 1.2, 8, 20, 7, max_batch=b, max_features=nf)
    ref_fe = ORBFeatureExtractor(w, h, 1.2, 8, 20, 7, max_batch=b, max_features=nf)
    cap = fe.geometry(nf).keypoint_capacity

    def call(ctx, img, budget):
        kp = torch.zeros((b, cap, 32), dtype=torch.uint8, device=dev)
        desc = torch.zeros((b, cap, 32), dtype=torch.uint8, device=dev)
        n = torch.zeros(b, dtype=torch.int32, device=dev)
        _lib.check(lib.orbgpu_extract_batch_device(ctx.handle, img.data_ptr(), w * h, w, b, budget, kp.data_ptr(), desc.data_ptr(),
                                                   cap, n.data_ptr()), ctx.handle)
        return kp, desc, n

    # (action, argument): "x" = extract imgs[arg] with the full budget, "x300" = with 300 keypoints
    script = [("hint", 1), ("x", 0), ("x", 1), ("x", 2), ("x", 0), ("x", 1), ("hint", 0), ("x", 2), ("x", 0), ("hint", 1),
              ("x", 1), ("x", 2), ("sync", 0), ("x", 0), ("x300", 1), ("x", 2), ("trim", 0), ("x", 1), ("x", 0), ("timing", 1),
              ("x", 2), ("timing", 0), ("x", 1), ("x", 2), ("x", 0)]
    got, want = [], []
    torch.cuda.synchronize()
    for act, arg in script:
        if act == "hint":
            _lib.check(lib.orbgpu_set_input_ready(fe.handle, arg, None), fe.handle)
        elif act == "sync":
            _lib.check(lib.orbgpu_synchronize(fe.handle), fe.handle)
        elif act == "trim":
            _lib.check(lib.orbgpu_trim(fe.handle), fe.handle)
        elif act == "timing":
            _lib.check(lib.orbgpu_set_stage_timing(fe.handle, arg), fe.handle)
        else:
            budget = 300 if act == "x300" else nf
            got.append(call(fe, imgs[arg], budget))
            want.append(call(ref_fe, imgs[arg], budget))
    _lib.check(lib.orbgpu_synchronize(fe.handle), fe.handle)
    _lib.check(lib.orbgpu_synchronize(ref_fe.handle), ref_fe.handle)
    assert len(got) == sum(1 for a, _ in script if a.startswith("x"))
    for i, (g_, w_) in enumerate(zip(got, want)):
        n = w_[2].cpu().numpy()
        assert torch.equal(g_[2], w_[2]), f"call {i}: counts"
        gk, wk, gd, wd = g_[0].cpu().numpy(), w_[0].cpu().numpy(), g_[1].cpu().numpy(), w_[1].cpu().numpy()
        for f in range(b):
            assert np.array_equal(gk[f, :n[f]], wk[f, :n[f]]) and np.array_equal(gd[f, :n[f]], wd[f, :n[f]]), f"call {i}, frame {f}"
    # the stage dump reads the working set of the last call
    da, db = fe.debug_dump(3), ref_fe.debug_dump(3)
    for key in ("pyramid", "score", "blurred", "cand_x", "cand_y", "cand_score"):
        assert np.array_equal(da[key], db[key]), key
    fe.close()
    ref_fe.close()


@pytest.mark.gpu
def test_one_frame_graph_survives_working_set_changes():
    """A batch context's one-frame host call replays a captured graph that holds the addresses of the working set of the
    time of its capture.  Calls under the input-ready hint then switch working sets and orbgpu_trim frees one of them:
    the next one-frame call must capture again, not replay addresses that are gone."""
    import torch
    from orb_slam3_amd import _lib
    from orb_slam3_amd.features import Features, ORBFeatureExtractor
    lib = _lib.load()
    w, h, nf, b = 640, 480, 400, 8
    seq = synth.sequence(31, b, h, w)
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(np.ascontiguousarray(seq)).to(dev)
    fe = ORBFeatureExtractor(w, h, 1.2, 8, 20, 7, max_batch=b, max_features=nf)
    cap = fe.geometry(nf).keypoint_capacity

    def one(frame):
        ft = Features(w, h)
        assert fe.Extract(seq[frame], ft, nf) == 0
        return ft

    first = one(3)                       # captures the graph on working set 0
    kp = torch.zeros((b, cap, 32), dtype=torch.uint8, device=dev)
    desc = torch.zeros((b, cap, 32), dtype=torch.uint8, device=dev)
    n = torch.zeros(b, dtype=torch.int32, device=dev)
    _lib.check(lib.orbgpu_set_input_ready(fe.handle, 1, None), fe.handle)
    for _ in range(3):                   # an odd number of switches: the context ends on the second set
        _lib.check(lib.orbgpu_extract_batch_device(fe.handle, d_img.data_ptr(), w * h, w, b, nf, kp.data_ptr(), desc.data_ptr(),
                                                   cap, n.data_ptr()), fe.handle)
    again = one(3)                       # both sets alive, the context on the other one: captured again
    dump = fe.debug_dump(0)              # ... so the stage dump (which reads the current set) sees this call's pyramid
    assert np.array_equal(dump["pyramid"][:w * h].reshape(h, w), seq[3])
    _lib.check(lib.orbgpu_set_input_ready(fe.handle, 0, None), fe.handle)
    _lib.check(lib.orbgpu_trim(fe.handle), fe.handle)   # frees the set the graph was captured on
    after = one(3)
    other = one(5)
    for ft in (again, after):
        assert np.array_equal(first.keypoints, ft.keypoints) and np.array_equal(first.descriptors, ft.descriptors)
    nn = int(n[5].item())
    assert len(other.keypoints) == nn and np.array_equal(other.descriptors, desc[5, :nn].cpu().numpy())
    fe.close()
