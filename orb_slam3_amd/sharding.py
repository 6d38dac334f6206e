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


def gather_keyframe_descriptors(local_desc, local_n, group=None):
    """All-gather ragged keyframe descriptor shards.

    local_desc: [K_local, cap, 32] uint8, local_n: [K_local] int32 (keypoints per keyframe), on the rank's device.
    Returns (all_desc [K_total, cap, 32], all_n [K_total], row_offset) where this rank's keyframes are rows
    row_offset .. row_offset + K_local of the gathered set, in rank order (= sequence order for block sharding).
    One all_gather of equal-sized padded shards (RCCL needs equal counts) plus one of the shard sizes.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_desc, local_n, 0
    rank = dist.get_rank(group)
    dev = local_desc.device
    k_local = torch.tensor([local_desc.shape[0]], dtype=torch.int64, device=dev)
    ks = [torch.zeros_like(k_local) for _ in range(world)]
    dist.all_gather(ks, k_local, group=group)
    ks = [int(k.item()) for k in ks]
    k_max = max(ks)
    cap = local_desc.shape[1]
    pad_desc = torch.zeros((k_max, cap, 32), dtype=torch.uint8, device=dev)
    pad_n = torch.zeros(k_max, dtype=torch.int32, device=dev)
    pad_desc[:ks[rank]] = local_desc
    pad_n[:ks[rank]] = local_n
    out_desc = torch.empty((world * k_max, cap, 32), dtype=torch.uint8, device=dev)
    out_n = torch.empty(world * k_max, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out_desc, pad_desc, group=group)
    dist.all_gather_into_tensor(out_n, pad_n, group=group)
    keep = torch.cat([torch.arange(r * k_max, r * k_max + ks[r], device=dev) for r in range(world)])
    return out_desc[keep].contiguous(), out_n[keep].contiguous(), sum(ks[:rank])


def allpairs_plan(row_offset, k_local, k_total):
    """(to_row, from_row) keyframe pairs this rank evaluates: its rows against every other keyframe."""
    return [(i, j) for i in range(row_offset, row_offset + k_local) for j in range(k_total) if j != i]
