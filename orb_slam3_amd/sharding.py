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


def gather_keyframe_rows(tensors, group=None, _force_collective=False):
    """All-gather ragged per-keyframe shards.  tensors: sequence of [K_local, ...] tensors sharing K_local.

    Returns (gathered list of [K_total, ...], row_offset): this rank's keyframes are rows row_offset ..
    row_offset + K_local of the gathered set, in rank order (= sequence order for block sharding).
    One all_gather of the shard sizes, then one all_gather_into_tensor of equal-sized padded shards per tensor
    (RCCL needs equal counts).  _force_collective runs the collectives on a single rank too (a test of the backend)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (_force_collective and dist.is_initialized()):
        return list(tensors), 0
    rank = dist.get_rank(group)
    dev = tensors[0].device
    host = _needs_host_staging(tensors[0], group)
    cdev = torch.device("cpu") if host else dev
    k_local = torch.tensor([tensors[0].shape[0]], dtype=torch.int64, device=cdev)
    ks = [torch.zeros_like(k_local) for _ in range(world)]
    dist.all_gather(ks, k_local, group=group)
    ks = [int(k.item()) for k in ks]
    k_max = max(max(ks), 1)
    keep = torch.cat([torch.arange(r * k_max, r * k_max + ks[r], device=cdev) for r in range(world)])
    out = []
    for t in tensors:
        pad = torch.zeros((k_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=cdev)
        pad[:ks[rank]] = t.to(cdev)
        full = torch.empty((world * k_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=cdev)
        dist.all_gather_into_tensor(full, pad, group=group)
        out.append(full[keep].contiguous().to(dev))
    return out, sum(ks[:rank])


def gather_keyframe_descriptors(local_desc, local_n, group=None):
    """local_desc: [K_local, cap, 32] uint8, local_n: [K_local] int32 -> (all_desc, all_n, row_offset)."""
    (desc, n), off = gather_keyframe_rows([local_desc, local_n], group)
    return desc, n, off


def match_keyframes_allpairs(extractor, local_kp, local_desc, local_n, threshold=50, ratio=0.9, check_orientation=True,
                             want_lists=False, group=None):
    """Cross-shard all-pairs keyframe matching: ONE exchange step -- the all-gather of the keyframe
    descriptor shards (and of the keypoint records when the rotation filter is on) -- then every rank matches ITS
    keyframes (rows) against ALL keyframes (columns) with orbgpu_match_allpairs_device.  Results stay sharded by row.

    local_kp [K_local, cap, 32] uint8 (orbgpu_keypoint records), local_desc [K_local, cap, 32] uint8,
    local_n [K_local] int32: device tensors laid out as orbgpu_extract_batch_device writes them.
    Returns (n_matches [K_local, K_total] int32, match [K_local, K_total, cap] int32 or None, row_offset)."""
    from . import _lib
    lib = _lib.load()
    if not local_desc.is_cuda:
        raise _lib.OrbGpuError(_lib.E_INVALID, "match_keyframes_allpairs needs device tensors (there is no CPU path)")
    send = [local_desc, local_n] + ([local_kp] if check_orientation else [])
    got, row_offset = gather_keyframe_rows(send, group)
    all_desc, all_n = got[0], got[1]
    all_kp = got[2] if check_orientation else None
    k_local, cap = local_desc.shape[0], local_desc.shape[1]
    k_total = all_desc.shape[0]
    dev = local_desc.device
    n_matches = torch.zeros((k_local, k_total), dtype=torch.int32, device=dev)
    match = torch.empty((k_local, k_total, cap), dtype=torch.int32, device=dev) if want_lists else None
    # the library works on its own stream: make the gathered tensors visible to it, and its results to torch
    torch.cuda.current_stream(dev).synchronize()
    rc = lib.orbgpu_match_allpairs_device(
        extractor.handle, local_kp.data_ptr() if check_orientation else None, local_desc.data_ptr(), local_n.data_ptr(),
        k_local, all_kp.data_ptr() if check_orientation else None, all_desc.data_ptr(), all_n.data_ptr(), k_total, cap,
        row_offset, int(threshold), float(ratio), 1 if check_orientation else 0, n_matches.data_ptr(),
        match.data_ptr() if want_lists else None)
    _lib.check(rc, extractor.handle)
    _lib.check(lib.orbgpu_synchronize(extractor.handle), extractor.handle)
    return n_matches, match, row_offset


def allpairs_plan(row_offset, k_local, k_total):
    """(to_row, from_row) keyframe pairs this rank evaluates: its rows against every other keyframe."""
    return [(i, j) for i in range(row_offset, row_offset + k_local) for j in range(k_total) if j != i]
