"""Frame sharding for the offline workloads: one process per GPU, contiguous blocks of frames per
rank (so frame t and t-1 stay on one rank; one halo frame at each block start), no data-path collective for
extraction and frame-to-frame linking, and ONE all-gather of keyframe descriptor shards for cross-shard all-pairs
keyframe matching.  Works with any torch.distributed backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the
CPU tests.  The reference has no counterpart (single process); matching semantics are those of
SNNMatcher (second_nearest_neighbor_matcher.hpp:30-80) with no candidate restriction.
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class FrameBlock:
    start: int        # first frame this rank extracts and owns
    stop: int         # one past the last owned frame
    halo_start: int   # first frame this rank must extract to link its first owned frame (start-1, or start at 0)

    @property
    def n_owned(self):
        return self.stop - self.start

    @property
    def n_extract(self):
        return self.stop - self.halo_start


def frame_block(n_frames, world_size, rank):
    """Contiguous block partition: the first n_frames % world_size ranks get one extra frame."""
    base, rem = divmod(n_frames, world_size)
    start = rank * base + min(rank, rem)
    stop = start + base + (1 if rank < rem else 0)
    halo = start - 1 if (start > 0 and stop > start) else start
    return FrameBlock(start, stop, halo)


def keyframe_ids(block, every):
    """Global ids of the keyframes a rank owns (every `every`-th frame of the sequence)."""
    first = ((block.start + every - 1) // every) * every
    return list(range(first, block.stop, every))


def _needs_host_staging(t, group):
    """gloo has no all_gather for device tensors: the CPU/gloo rehearsal stages through host memory."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _gather_padded(tensors, group=None, on_sizes_known=None):
    """All-gather ragged per-keyframe shards in their PADDED form: one all_gather of the shard sizes, then one
    all_gather_into_tensor of equal-sized shards per tensor (RCCL needs equal counts; every shard is padded to the
    largest).  Returns (fulls, ks): fulls[i] is [world * k_max, ...] with rank r's rows at r * k_max .. r * k_max + ks[r]
    and zeros in the padding.  on_sizes_known (optional) runs between the size exchange and the payload collectives."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = tensors[0].device
    host = _needs_host_staging(tensors[0], group)
    cdev = torch.device("cpu") if host else dev
    k_local = torch.tensor([tensors[0].shape[0]], dtype=torch.int64, device=cdev)
    ks = [torch.zeros_like(k_local) for _ in range(world)]
    dist.all_gather(ks, k_local, group=group)
    ks = [int(k.item()) for k in ks]
    k_max = max(max(ks), 1)
    if on_sizes_known is not None:
        on_sizes_known(ks, k_max)
    fulls = []
    for t in tensors:
        pad = torch.zeros((k_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=cdev)
        pad[:ks[rank]] = t.to(cdev)
        full = torch.empty((world * k_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=cdev)
        dist.all_gather_into_tensor(full, pad, group=group)
        fulls.append(full.to(dev))
    return fulls, ks


def gather_keyframe_rows(tensors, group=None, _force_collective=False):
    """All-gather ragged per-keyframe shards.  tensors: sequence of [K_local, ...] tensors sharing K_local.

    Returns (gathered list of [K_total, ...], row_offset): this rank's keyframes are rows row_offset ..
    row_offset + K_local of the gathered set, in rank order (= sequence order for block sharding).
    _force_collective runs the collectives on a single rank too (a test of the backend)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (_force_collective and dist.is_initialized()):
        return list(tensors), 0
    rank = dist.get_rank(group)
    fulls, ks = _gather_padded(tensors, group)
    k_max = max(max(ks), 1)
    keep = torch.cat([torch.arange(r * k_max, r * k_max + ks[r], device=fulls[0].device) for r in range(world)])
    return [f[keep].contiguous() for f in fulls], sum(ks[:rank])


def gather_keyframe_descriptors(local_desc, local_n, group=None):
    """local_desc: [K_local, cap, 32] uint8, local_n: [K_local] int32 -> (all_desc, all_n, row_offset)."""
    (desc, n), off = gather_keyframe_rows([local_desc, local_n], group)
    return desc, n, off


def keypoint_angles(kp):
    """[K, cap, 32] uint8 orbgpu_keypoint records -> [K, cap] float32 angles (bytes 16..19 of a record): all the rotation
    filter reads of a keypoint, and 8 x fewer bytes over xGMI than the records."""
    return kp.view(torch.float32)[..., 4].contiguous()


class StreamBridge:
    """Orders the library's stream against torch's streams with the C ABI's orbgpu_wait_event / orbgpu_record_event (no
    device-wide synchronisation).  The torch events are created once and re-recorded."""

    def __init__(self, extractor):
        from . import _lib
        self._lib, self._mod, self._h = _lib.load(), _lib, extractor.handle
        self._to_lib = torch.cuda.Event()
        self._from_lib = torch.cuda.Event()
        self._from_lib.record()          # events are created by their first record: the library needs a live handle

    def lib_waits_for_torch(self, stream=None):
        """work enqueued by the library from now on starts after what `stream` (default: torch's current) holds now"""
        self._to_lib.record(stream if stream is not None else torch.cuda.current_stream())
        self._mod.check(self._lib.orbgpu_wait_event(self._h, self._to_lib.cuda_event), self._h)

    def torch_waits_for_lib(self, stream=None):
        """work enqueued on `stream` (default: torch's current) from now on starts after the library's work so far"""
        self._mod.check(self._lib.orbgpu_record_event(self._h, self._from_lib.cuda_event), self._h)
        (stream if stream is not None else torch.cuda.current_stream()).wait_event(self._from_lib)


def match_keyframes_allpairs(extractor, local_kp, local_desc, local_n, threshold=50, ratio=0.9, check_orientation=True,
                             want_lists=False, group=None):
    """Cross-shard all-pairs keyframe matching: ONE exchange step -- the all-gather of the keyframe
    descriptor shards (plus, when the rotation filter is on, a float angle per keypoint: 36 instead of 64 bytes per
    keypoint cross xGMI) -- then every rank matches ITS keyframes (rows) against ALL keyframes (columns) with
    orbgpu_match_allpairs_angles_device.  The rows x local-columns block needs no exchange and is enqueued first, so the
    GPU matches while the collective is in flight; the remote columns follow as two calls (the shards before and after
    this rank's own).  Results stay sharded by row.  No device-wide synchronisation: the library's stream and torch's are
    ordered with orbgpu_wait_event / orbgpu_record_event.

    local_kp [K_local, cap, 32] uint8 (orbgpu_keypoint records), local_desc [K_local, cap, 32] uint8,
    local_n [K_local] int32: device tensors laid out as orbgpu_extract_batch_device writes them.
    Returns (n_matches [K_local, K_total] int32, match [K_local, K_total, cap] int32 or None, row_offset); the results
    are ordered against torch's current stream."""
    from . import _lib
    lib = _lib.load()
    if not local_desc.is_cuda:
        raise _lib.OrbGpuError(_lib.E_INVALID, "match_keyframes_allpairs needs device tensors (there is no CPU path)")
    dev = local_desc.device
    k_local, cap = local_desc.shape[0], local_desc.shape[1]
    co = 1 if check_orientation else 0
    local_ang = keypoint_angles(local_kp) if check_orientation else None
    bridge = StreamBridge(extractor)
    bridge.lib_waits_for_torch()          # the inputs (and local_ang) were produced on torch's current stream

    def run(rows, cols, row_offset):
        """rows / cols: (angle or None, desc, n); returns (n_matches, match) of that block, enqueued on the library's stream"""
        nr, nc = rows[1].shape[0], cols[1].shape[0]
        nm = torch.zeros((nr, nc), dtype=torch.int32, device=dev)
        ml = torch.empty((nr, nc, cap), dtype=torch.int32, device=dev) if want_lists else None
        if nr and nc:
            bridge.lib_waits_for_torch()  # the outputs above were allocated / zeroed on torch's stream
            _lib.check(lib.orbgpu_match_allpairs_angles_device(
                extractor.handle, rows[0].data_ptr() if co else None, rows[1].data_ptr(), rows[2].data_ptr(), nr,
                cols[0].data_ptr() if co else None, cols[1].data_ptr(), cols[2].data_ptr(), nc, cap, row_offset,
                int(threshold), float(ratio), co, nm.data_ptr(), ml.data_ptr() if want_lists else None), extractor.handle)
        return nm, ml

    local = (local_ang, local_desc, local_n)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    nm_local, ml_local = run(local, local, 0)              # needs no exchange: in flight while the collective runs
    if world == 1:
        bridge.torch_waits_for_lib()
        return nm_local, ml_local, 0
    rank = dist.get_rank(group)
    send = [local_desc, local_n] + ([local_ang] if check_orientation else [])
    fulls, ks = _gather_padded(send, group)
    k_max = max(max(ks), 1)
    all_desc, all_n = fulls[0], fulls[1]
    all_ang = fulls[2] if check_orientation else None
    lo, hi = rank * k_max, (rank + 1) * k_max
    parts_n, parts_m = [], []
    for a, b in ((0, lo), (hi, world * k_max)):            # padded rows hold n = 0: they match nothing
        cols = (all_ang[a:b] if co else None, all_desc[a:b], all_n[a:b])
        nm, ml = run(local, cols, -1)
        parts_n.append(nm)
        parts_m.append(ml)
    bridge.torch_waits_for_lib()
    pad_n = torch.zeros((k_local, k_max - k_local), dtype=torch.int32, device=dev)
    n_padded = torch.cat([parts_n[0], nm_local, pad_n, parts_n[1]], dim=1)
    keep = torch.cat([torch.arange(r * k_max, r * k_max + ks[r], device=dev) for r in range(world)])
    n_matches = n_padded[:, keep].contiguous()
    match = None
    if want_lists:
        pad_m = torch.full((k_local, k_max - k_local, cap), -1, dtype=torch.int32, device=dev)
        match = torch.cat([parts_m[0], ml_local, pad_m, parts_m[1]], dim=1)[:, keep].contiguous()
    return n_matches, match, sum(ks[:rank])


def allpairs_plan(row_offset, k_local, k_total):
    """(to_row, from_row) keyframe pairs this rank evaluates: its rows against every other keyframe."""
    return [(i, j) for i in range(row_offset, row_offset + k_local) for j in range(k_total) if j != i]
