"""Frame sharding across the GPUs of one node.

Frames of a video carry no temporal state (src/can_swap_pipeline_e2e.py:223-283 reads only frame i's inputs and
the per-video ``source_id``), so the path shards by contiguous frame blocks with no data-path collective.  The
only communication is a one-time broadcast of the source identity (2 KB; every rank then derives T's modulated
weights locally) and a final gather of the uint8 output frames to rank 0.  Backend "nccl" is RCCL on ROCm
(xGMI); the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int):
    """Contiguous block [start, stop) of rank `rank`; blocks differ by at most one frame."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_identity(source_id: torch.Tensor, src: int = 0) -> torch.Tensor:
    """One-time broadcast of the 512-float identity embedding from rank `src` (in place)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(source_id, src=src)
    return source_id


def gather_frames(local: torch.Tensor, n_frames: int, dst: int = 0):
    """Gather per-rank uint8 frame blocks [n_local, H, W, 3] to rank `dst` in frame order.
    Returns the [n_frames, H, W, 3] tensor on `dst`, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_range(n_frames, r, world) for r in range(world)]
    nmax = max(b - a for a, b in counts)
    pad = local
    if local.shape[0] < nmax:   # equal-sized pieces keep this one collective
        pad = torch.cat([local, local.new_zeros((nmax - local.shape[0],) + tuple(local.shape[1:]))], 0)
    pieces = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), pieces, dst=dst)
    if rank != dst:
        return None
    return torch.cat([p[: b - a] for p, (a, b) in zip(pieces, counts)], 0)
