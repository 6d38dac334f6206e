"""Offline sequence runner (BASELINE.json configs[4], SURVEY 8d input 5 / 8e): the reference's sequence loop --
RunDataset, main.cpp:334-411: per frame one MonocularFrame construction (= IFeatureExtractor::Extract,
factories/feature_handler_factory.cpp:50) and, through Tracker::Track, the link to the previous frame
(MonocularFrame::ComputeMatchesForLinking, frame/monocular/monocular_frame.cpp:173-200) -- restated for a
frame-sharded offline job on MI355X:

  * a rank owns a contiguous block of the sequence (sharding.frame_block) and walks it in batches of the extractor's
    max_batch frames: upload -> orbgpu_extract_batch_device -> orbgpu_match_frames_device (t -> t-1);
  * links cross batch boundaries (the last frame of a batch is carried on the device into slot 0 of the next batch's
    output arrays) and rank boundaries (a rank with a predecessor extracts ONE halo frame, start - 1, that it does not own);
  * every `keyframe_every`-th frame of the sequence is a keyframe: its keypoints and descriptors stay on the device for
    the cross-shard all-pairs matching (sharding.match_keyframes_allpairs: one all-gather, then rows x all columns);
  * uploads go through a ring of device image buffers on a copy stream of their own, ordered against the library's
    stream with orbgpu_wait_event / orbgpu_record_event only, so batch k+1 crosses PCIe while batch k is extracted;
  * given TWO extractors (two contexts of the library, each with its own working buffers and streams) the batches
    alternate between them, so that the tail of batch k (descriptors, link: gathers that wait on memory) runs beside the
    head of batch k+1 (pyramid, FAST: vector-ALU work) -- 9 % more frames per second with the frames resident; the
    carry of the last frame and the reuse of an output set are ordered with events between the two streams.

Nothing here computes on the CPU; torch is used for device memory, streams and the device-side bookkeeping copies.
"""
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib, synth
from .sharding import frame_block, keyframe_ids


# ---------------------------------------------------------------------------------------------- frame sources
class ProceduralSource:
    """Frame i generated on demand from (seed, i) (synth.ProceduralSequence): no dataset moves, any rank can produce any
    frame.  Host generation is slow (a numpy warp per frame), so this source is for tests and small runs."""

    def __init__(self, seed, height=480, width=640):
        self.seq = synth.ProceduralSequence(seed, height, width)
        self.height, self.width = height, width

    def runs(self, first, count):
        return None

    def fill(self, first, count, out):
        from concurrent.futures import ThreadPoolExecutor

        def one(k):
            out[k] = self.seq.frame(first + k)
        with ThreadPoolExecutor(8) as ex:      # numpy releases the GIL inside the warp's array operations
            list(ex.map(one, range(count)))


class PooledSource:
    """Frame i = pool[i % len(pool)], the pool (a short procedural sequence) held in PINNED host memory the way a
    decoder's output buffers would be: a batch is handed to the runner as slices of the pool and crosses PCIe straight
    from there (no host staging copy).  The throughput source of bench.py."""

    def __init__(self, seed, pool_frames=64, height=480, width=640, frames=None):
        if frames is None:
            seq = synth.ProceduralSequence(seed, height, width)
            frames = np.stack([seq.frame(i) for i in range(pool_frames)])
        self.height, self.width = int(frames.shape[1]), int(frames.shape[2])
        self.pool = torch.from_numpy(np.ascontiguousarray(frames)).pin_memory()
        self.frames = frames

    def index(self, i):
        return i % self.pool.shape[0]

    def runs(self, first, count):
        out, p = [], self.pool.shape[0]
        while count > 0:
            a = first % p
            n = min(count, p - a)
            out.append(self.pool[a:a + n])
            first += n
            count -= n
        return out

    def fill(self, first, count, out):
        for k in range(count):
            out[k] = self.frames[(first + k) % len(self.frames)]


class ResidentSource(PooledSource):
    """The same pool already resident in HBM: the 'inputs already on the device' leg of the measurement (what `value`
    of bench.py assumes); a batch is a device-to-device copy of pool slices."""

    def __init__(self, pooled, device, repeat=1):
        # repeat: the pool tiled that many times on the device (frame i is still pool frame i % len(pool)), so that a
        # whole batch can be one contiguous slice -- which the runner then extracts in place, without a copy
        self.height, self.width = pooled.height, pooled.width
        self.frames = pooled.frames
        self.pool = pooled.pool.to(device)
        if repeat > 1:
            self.pool = torch.cat([self.pool] * int(repeat))


# ---------------------------------------------------------------------------------------------- results
@dataclass
class SequenceResult:
    block: object                      # sharding.FrameBlock of this rank
    first_frame: int                   # global id of row 0 of the per-frame arrays (= block.start)
    n_keypoints: np.ndarray            # [n_owned] keypoints per owned frame
    n_links: np.ndarray                # [n_owned] links t -> t-1 that survive ratio, uniqueness and the rotation filter
    keyframe_ids: list                 # global ids of this rank's keyframes
    kf_kp: torch.Tensor                # [K_local, cap, 32] u8 orbgpu_keypoint records (device)
    kf_desc: torch.Tensor              # [K_local, cap, 32] u8 (device)
    kf_n: torch.Tensor                 # [K_local] i32 (device)
    capacity: int
    links: np.ndarray = None           # [n_owned, cap] 'from' index in frame t-1 per keypoint of frame t, or -1 (keep_links)
    keypoints: np.ndarray = None       # [n_owned, cap] KP_DTYPE (keep_features)
    descriptors: np.ndarray = None     # [n_owned, cap, 32] u8 (keep_features)
    timing: dict = field(default_factory=dict)


class _OutputSet:
    """what one context of the library works into: the output arrays of a batch (slot 0 = the frame carried over from
    the previous batch, slots 1 .. B = this batch), the link results, the context's stream as a torch stream (the
    device-side bookkeeping -- carry, keyframe gather, per-frame counts -- is enqueued on it between the library's calls,
    so the stream itself orders everything a set holds) and two events for the hand-over between two sets"""

    def __init__(self, lib, fe, B, cap, dev):
        self.fe, self.h = fe, fe.handle
        self.stream = torch.cuda.ExternalStream(int(lib.orbgpu_stream(fe.handle)), device=dev)
        self.d_kp = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
        self.d_desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
        self.d_n = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        self.d_match = torch.zeros((B, cap), dtype=torch.int32, device=dev)
        self.d_dist = torch.zeros((B, cap), dtype=torch.int32, device=dev)
        self.d_nm = torch.zeros(B, dtype=torch.int32, device=dev)
        self.ev_carry_in = torch.cuda.Event()    # this set's slot 0 holds the carry for its next batch
        self.ev_idle = torch.cuda.Event()        # this set's last batch is linked and gathered
        for e in (self.ev_carry_in, self.ev_idle):
            e.record(self.stream)


class SequenceRunner:
    """One rank's share of an offline sequence.  extractor: features.ORBFeatureExtractor created with max_batch = the
    batch size to walk in; source: a frame source (above).  ring: device image buffers in flight."""

    def __init__(self, extractor, source, feature_count, keyframe_every=10, window=100, threshold=50, ratio=0.9,
                 check_orientation=True, ring=4, copy_streams=1, input_ready=False):
        # one extractor, or a list of two of the same configuration: batches alternate between them
        self.fes = list(extractor) if isinstance(extractor, (list, tuple)) else [extractor]
        # input_ready: hand the upload event to the library as the batch's input event instead of making its stream wait
        # (orbgpu_set_input_ready: the call's head beside the previous batch's tail); pays when the extraction is the
        # bound, costs upload rate when PCIe is (one more stream on the process's 4 hardware queues)
        self.input_ready = bool(input_ready)
        self.fe, self.src = self.fes[0], source
        extractor = self.fe
        self.lib = _lib.load()
        self.nf = int(feature_count)
        self.every = int(keyframe_every)
        self.link_args = (int(window), int(threshold), float(ratio), 1 if check_orientation else 0)
        self.B = extractor.max_batch
        self.H, self.W = source.height, source.width
        if (self.W, self.H) != (extractor.image_width_, extractor.image_height_):
            raise _lib.OrbGpuError(_lib.E_INVALID, "frame source and extractor disagree on the image size")
        self.cap = extractor.geometry(self.nf).keypoint_capacity
        for fe in self.fes[1:]:
            if (fe.max_batch, fe.image_width_, fe.image_height_, fe._p.device, fe.geometry(self.nf).keypoint_capacity) != \
                    (self.B, self.W, self.H, extractor._p.device, self.cap):
                raise _lib.OrbGpuError(_lib.E_INVALID, "the extractors of a sequence runner must be configured alike")
        self.dev = torch.device("cuda", extractor._p.device)
        self.ring = max(2, int(ring))
        B, cap, dev = self.B, self.cap, self.dev
        with torch.cuda.device(self.dev):
            self.n_copy = max(1, int(copy_streams))
            self.copy_streams = [torch.cuda.Stream(self.dev) for _ in range(self.n_copy)]
            self.copy_stream = self.copy_streams[0]
            self.ev_join = [torch.cuda.Event() for _ in range(self.n_copy - 1)]
            self.d_img = [torch.empty((B, self.H, self.W), dtype=torch.uint8, device=dev) for _ in range(self.ring)]
            self.h_img = None                      # pinned staging ring, only for sources without pinned runs
            self.batch_in = list(self.d_img)       # what the extraction of ring slot r reads: the slot, or the source's own memory
            self.ev_up = [torch.cuda.Event() for _ in range(self.ring)]     # upload of ring slot r has landed
            self.ev_free = [torch.cuda.Event() for _ in range(self.ring)]   # extraction of ring slot r has read it
            for e in self.ev_up + self.ev_free:
                e.record(self.copy_stream)         # created by their first record (the library needs live handles)
            self.d_to = torch.arange(1, B + 1, dtype=torch.int32, device=dev)
            self.d_from = torch.arange(0, B, dtype=torch.int32, device=dev)
            self.sets = [_OutputSet(self.lib, fe, B, cap, dev) for fe in self.fes]
            torch.cuda.synchronize(self.dev)

    def _check(self, rc, handle=None):
        _lib.check(rc, handle if handle is not None else self.fe.handle)

    def _upload(self, k, first, count):
        """batch k (frames first .. first + count - 1) into ring slot k % ring, on the copy stream(s)"""
        r = k % self.ring
        cs = self.copy_stream
        cs.wait_event(self.ev_free[r])             # the extraction that last read this slot has finished with it
        runs = self.src.runs(first, count)
        if runs is not None and len(runs) == 1 and runs[0].is_cuda and runs[0].is_contiguous():
            # the batch is one slice of device memory: extracted where it lies
            self.batch_in[r] = runs[0]
            self.ev_up[r].record(cs)
            return 0
        self.batch_in[r] = self.d_img[r]
        if runs is None:                           # host generator: fill a pinned staging slot, then one copy
            if self.h_img is None:
                self.h_img = [torch.empty((self.B, self.H, self.W), dtype=torch.uint8).pin_memory()
                              for _ in range(self.ring)]
            self.ev_up[r].synchronize()            # the previous upload from this pinned slot has left the host
            self.src.fill(first, count, self.h_img[r].numpy())
            runs = [self.h_img[r][:count]]
        # with more than one copy stream the pieces of a batch alternate between them (measured on MI355X: one stream
        # reaches 50 GB/s of upload, two to four 34-42 -- the default is one)
        pieces, o = [], 0
        for t in runs:
            n, step = t.shape[0], max(1, (t.shape[0] + self.n_copy - 1) // self.n_copy) if len(runs) < self.n_copy else t.shape[0]
            for a in range(0, n, step):
                b = min(a + step, n)
                pieces.append((o + a, t[a:b]))
            o += n
        extra = self.copy_streams[1:]
        for q, st in enumerate(extra):
            st.wait_event(self.ev_free[r])
        for i, (dst0, t) in enumerate(pieces):
            st = self.copy_streams[i % self.n_copy]
            with torch.cuda.stream(st):
                self.d_img[r][dst0:dst0 + t.shape[0]].copy_(t, non_blocking=True)
        for q, st in enumerate(extra):             # join the side copy streams into the first one
            self.ev_join[q].record(st)
            cs.wait_event(self.ev_join[q])
        self.ev_up[r].record(cs)
        return count * self.H * self.W

    def run(self, n_frames, rank=0, world=1, keep_links=False, keep_features=False):
        """Extract + link this rank's block of an n_frames sequence; returns a SequenceResult (per-frame arrays on the
        host, keyframes on the device).  Host-side work per batch is enqueue-only; the one wait is at the end."""
        lib = self.lib
        B, cap, dev, nf = self.B, self.cap, self.dev, self.nf
        blk = frame_block(n_frames, world, rank)
        kf_ids = keyframe_ids(blk, self.every)
        n_ext, n_own = blk.n_extract, blk.n_owned
        halo = blk.start - blk.halo_start                      # 0 or 1 frames extracted but not owned
        with torch.cuda.device(dev):
            all_n = torch.zeros(max(n_ext, 1), dtype=torch.int32, device=dev)
            all_nm = torch.zeros(max(n_ext, 1), dtype=torch.int32, device=dev)
            K = len(kf_ids)
            kf_kp = torch.zeros((K, cap, 32), dtype=torch.uint8, device=dev)
            kf_desc = torch.zeros((K, cap, 32), dtype=torch.uint8, device=dev)
            kf_n = torch.zeros(K, dtype=torch.int32, device=dev)
            all_links = torch.full((n_ext, cap), -1, dtype=torch.int32, device=dev) if keep_links else None
            all_kp = torch.zeros((n_ext, cap, 32), dtype=torch.uint8, device=dev) if keep_features else None
            all_desc = torch.zeros((n_ext, cap, 32), dtype=torch.uint8, device=dev) if keep_features else None
            torch.cuda.synchronize(dev)
        batches = [(s, min(s + B, blk.stop)) for s in range(blk.halo_start, blk.stop, B)]
        window, th, ratio, co = self.link_args
        t0 = time.perf_counter()
        bytes_up = 0
        ahead = min(self.ring - 1, len(batches))               # uploads in flight ahead of the extraction
        for k in range(ahead):
            bytes_up += self._upload(k, batches[k][0], batches[k][1] - batches[k][0])
        kf_pos = 0
        for k, (g0, g1) in enumerate(batches):
            c = g1 - g0
            r = k % self.ring
            S = self.sets[k % len(self.sets)]
            h = S.h
            if k + ahead < len(batches):
                nb = batches[k + ahead]
                bytes_up += self._upload(k + ahead, nb[0], nb[1] - nb[0])
            # the batch's images are complete when its upload event fires: the library may start the call's pyramid and
            # FAST kernels then, beside the previous batch's descriptor and link kernels (orbgpu_set_input_ready)
            if self.input_ready:
                self._check(lib.orbgpu_set_input_ready(h, 1, self.ev_up[r].cuda_event), h)
            else:
                self._check(lib.orbgpu_wait_event(h, self.ev_up[r].cuda_event), h)
            self._check(lib.orbgpu_extract_batch_device(h, self.batch_in[r].data_ptr(), self.H * self.W, self.W, c, nf,
                                                        S.d_kp[1].data_ptr(), S.d_desc[1].data_ptr(), cap,
                                                        S.d_n[1:].data_ptr()), h)
            self._check(lib.orbgpu_record_event(h, self.ev_free[r].cuda_event), h)
            two = len(self.sets) > 1
            if two and k + 1 < len(batches):
                # carry across the two sets: the last frame of this batch is 'from' of the next batch's first link, which
                # the OTHER context computes.  Pushed from this set's stream right behind the extraction (no stream of
                # its own: every stream that waits shares one of the process's 4 hardware queues with others and holds
                # them up); the other set's slot 0 is free once its last batch is linked and gathered
                N = self.sets[(k + 1) % len(self.sets)]
                with torch.cuda.stream(S.stream):
                    S.stream.wait_event(N.ev_idle)
                    N.d_kp[0].copy_(S.d_kp[c])
                    N.d_desc[0].copy_(S.d_desc[c])
                    N.d_n[0:1].copy_(S.d_n[c:c + 1])
                    N.ev_carry_in.record(S.stream)
            if two and k > 0:
                S.stream.wait_event(S.ev_carry_in)
            # links: slot i -> slot i - 1.  The first batch has nothing in slot 0 (frame 0 of the sequence has no
            # predecessor; a halo frame is not owned), so its first pair is skipped
            p0 = 1 if k == 0 else 0
            if c - p0 > 0:
                self._check(lib.orbgpu_match_frames_device(
                    h, S.d_kp.data_ptr(), S.d_desc.data_ptr(), S.d_n.data_ptr(), cap, self.d_to[p0:].data_ptr(),
                    self.d_from[p0:].data_ptr(), c - p0, window, th, ratio, co, S.d_match[p0:].data_ptr(),
                    S.d_dist[p0:].data_ptr(), S.d_nm[p0:].data_ptr()), h)
            o = g0 - blk.halo_start
            with torch.cuda.stream(S.stream):
                all_n[o:o + c].copy_(S.d_n[1:c + 1])
                if c - p0 > 0:
                    all_nm[o + p0:o + c].copy_(S.d_nm[p0:c])
                    if keep_links:
                        all_links[o + p0:o + c].copy_(S.d_match[p0:c])
                if keep_features:
                    all_kp[o:o + c].copy_(S.d_kp[1:c + 1])
                    all_desc[o:o + c].copy_(S.d_desc[1:c + 1])
                k1 = kf_pos
                while k1 < K and kf_ids[k1] < g1:
                    k1 += 1
                if k1 > kf_pos:                                # keyframes of this batch: slots of a regular stride
                    s0 = kf_ids[kf_pos] - g0 + 1
                    sl = slice(s0, s0 + (k1 - kf_pos - 1) * self.every + 1, self.every)
                    kf_kp[kf_pos:k1].copy_(S.d_kp[sl])
                    kf_desc[kf_pos:k1].copy_(S.d_desc[sl])
                    kf_n[kf_pos:k1].copy_(S.d_n[sl])
                    kf_pos = k1
                if len(self.sets) == 1:
                    # carry inside the one set: the last frame of this batch is 'from' of the next batch's first link
                    S.d_kp[0].copy_(S.d_kp[c])
                    S.d_desc[0].copy_(S.d_desc[c])
                    S.d_n[0:1].copy_(S.d_n[c:c + 1])
                S.ev_idle.record(S.stream)
        t_enq = time.perf_counter() - t0
        for S in self.sets:
            self._check(lib.orbgpu_synchronize(S.h), S.h)      # also surfaces device-side capacity errors
            if self.input_ready:
                self._check(lib.orbgpu_set_input_ready(S.h, 0, None), S.h)
        for st in self.copy_streams:
            st.synchronize()
        t_all = time.perf_counter() - t0
        res = SequenceResult(
            block=blk, first_frame=blk.start, n_keypoints=all_n[halo:n_ext].cpu().numpy(),
            n_links=all_nm[halo:n_ext].cpu().numpy(), keyframe_ids=kf_ids, kf_kp=kf_kp, kf_desc=kf_desc, kf_n=kf_n,
            capacity=cap,
            links=all_links[halo:].cpu().numpy() if keep_links else None,
            keypoints=all_kp[halo:].cpu().numpy().view(_lib.KP_DTYPE).reshape(n_own, cap) if keep_features else None,
            descriptors=all_desc[halo:].cpu().numpy() if keep_features else None,
            timing={"seconds": t_all, "enqueue_seconds": t_enq, "frames_extracted": n_ext, "frames_owned": n_own,
                    "batches": len(batches), "bytes_uploaded": bytes_up,
                    "frames_per_s": n_ext / t_all if t_all > 0 else 0.0,
                    "upload_GBps": bytes_up / t_all / 1e9 if t_all > 0 else 0.0})
        return res

    def match_keyframes(self, result, threshold=50, ratio=0.9, check_orientation=True, row_block=256, group=None):
        """All-pairs keyframe matching of a finished run: this rank's keyframes against every rank's
        (sharding.match_keyframes_allpairs, in blocks of rows so that a launch stays bounded).
        Returns (n_matches [K_local, K_total] int32 on the device, row_offset)."""
        from .sharding import match_keyframes_allpairs
        K = result.kf_desc.shape[0]
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        if world > 1 or K <= row_block:
            nm, _, off = match_keyframes_allpairs(self.fe, result.kf_kp, result.kf_desc, result.kf_n, threshold, ratio,
                                                  check_orientation, False, group)
            return nm, off
        # one rank: rows in blocks against all columns (row_offset names the block's first column)
        from .sharding import StreamBridge, keypoint_angles
        bridge = StreamBridge(self.fe)
        ang = keypoint_angles(result.kf_kp)
        nm = torch.zeros((K, K), dtype=torch.int32, device=self.dev)
        bridge.lib_waits_for_torch()
        co = 1 if check_orientation else 0
        for r0 in range(0, K, row_block):
            r1 = min(r0 + row_block, K)
            self._check(self.lib.orbgpu_match_allpairs_angles_device(
                self.fe.handle, ang[r0:].data_ptr(), result.kf_desc[r0:].data_ptr(), result.kf_n[r0:].data_ptr(), r1 - r0,
                ang.data_ptr(), result.kf_desc.data_ptr(), result.kf_n.data_ptr(), K, result.capacity, r0, int(threshold),
                float(ratio), co, nm[r0:].data_ptr(), None))
        bridge.torch_waits_for_lib()
        return nm, 0
