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


def broadcast_identity(source_id: torch.Tensor, src: int = 0, group=None, single_rank: bool = False) -> torch.Tensor:
    """One-time broadcast of the 512-float identity embedding from (global) rank `src` (in place), over `group` if given.
    single_rank: issue the collective on a one-rank communicator too (bench.py --force-dist: the RCCL call path at N = 1)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or single_rank):
        dist.broadcast(source_id, src=src, group=group)
    return source_id


def stream_groups(world: int, n_streams: int):
    """BASELINE configs[4] placement: `n_streams` concurrent videos on `world` ranks.

    world >= n_streams: stream s owns the contiguous ranks {s*g, ..., s*g + g - 1}, g = world / n_streams (8 GPUs, 4 streams:
    stream s -> GPU pair {2s, 2s+1}); its frames shard over those ranks, its identity lives in identity slot 0 of each of them and
    its frames are gathered to the first rank of the group over a per-stream sub-communicator.
    world < n_streams: every rank hosts n_streams / world whole streams (stream s on rank s mod world), one identity slot per hosted
    stream, frames of the hosted streams interleaved in one launch (the per-sample modulated convolution of
    adaptive_modulate.py:157-167).
    Returns a list of rank lists, one per stream."""
    if world < 1 or n_streams < 1:
        raise ValueError("world and n_streams must be positive")
    if world >= n_streams:
        if world % n_streams:
            raise ValueError(f"{world} ranks do not split evenly over {n_streams} streams")
        g = world // n_streams
        return [list(range(s * g, (s + 1) * g)) for s in range(n_streams)]
    if n_streams % world:
        raise ValueError(f"{n_streams} streams do not split evenly over {world} ranks")
    return [[s % world] for s in range(n_streams)]


def streams_of_rank(groups, rank: int):
    """[(stream, position of `rank` inside the stream's group, identity slot on this rank)] for every stream `rank` serves."""
    mine = [s for s, g in enumerate(groups) if rank in g]
    return [(s, groups[s].index(rank), k) for k, s in enumerate(mine)]


def make_stream_comms(groups, min_ranks: int = 2):
    """One sub-communicator per multi-rank stream (every rank must call this with the same `groups`: dist.new_group is
    collective over the default group).  Entry s is None for streams with fewer than `min_ranks` ranks (default: single-rank
    streams need no communicator; bench.py --force-dist passes 1) or without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return [None] * len(groups)
    return [dist.new_group(ranks=g) if len(g) >= min_ranks else None for g in groups]


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


def equal_chunks(n_local: int, max_chunk: int):
    """Cut a rank's n_local frames into the fewest launches of at most max_chunk frames, all (but possibly the last) of the SAME size:
    150 frames at a 64-frame maximum run as 3 x 50, not 64 + 64 + 22 (odd last batches are slower per frame and make the ranks' gathers
    ragged).  Returns (chunk size, [(start, count), ...])."""
    if n_local <= 0:
        return max(1, min(max_chunk, 1)), []
    n = (n_local + max_chunk - 1) // max_chunk
    size = (n_local + n - 1) // n
    return size, [(t0, min(size, n_local - t0)) for t0 in range(0, n_local, size)]


class ChunkedFrameGather:
    """uint8 output frames leave every rank in chunks as soon as a chunk is finished: each chunk is one asynchronous gather
    to `dst` (one hop over xGMI under RCCL), overlapped with the computation of the next chunk (786 KB
    per frame, 118 MB per rank for 1200 frames over 8 GPUs).  All ranks issue the same number of equally sized collectives
    (short or missing chunks are padded); finish() waits for them and returns the frames in frame order on `dst`.

    The object is built ONCE per job shape and reused for every pass over the video (reset() between passes): the receive buffer on
    `dst` (943 MB for 1200 frames) is allocated here, never inside a timed region, and when every rank's share is a whole number of
    chunks (1200 frames over 8 ranks in chunks of 50) the chunks land directly in frame order - finish() returns a view, no copy.

    n_frames: total frames of the job, sharded by shard_range(); chunk: frames per collective (equal_chunks() of the largest share).

    Aliasing: on the exact path finish() returns a VIEW of the persistent receive buffer; the next pass (reset() + push()) overwrites
    it in place.  A caller that keeps the frames of pass i beyond the start of pass i + 1 asks for finish(copy=True); the ragged path
    always returns a fresh tensor.
    """

    def __init__(self, n_frames: int, chunk: int, frame_shape=(512, 512, 3), device="cpu", dst: int = 0, group=None,
                 single_rank: bool = False):
        """group: sub-communicator of one stream (stream_groups / make_stream_comms); `dst` is then the rank INSIDE the group.
        single_rank: run the collectives on a one-rank communicator as well (bench.py --force-dist)."""
        self.on = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or single_rank)
        self.group = group
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.dst, self.chunk, self.n_frames = dst, chunk, n_frames
        self.dst_global = dist.get_global_rank(group, dst) if (self.on and group is not None) else dst
        self.counts = [shard_range(n_frames, r, self.world) for r in range(self.world)]
        self.n_local = self.counts[self.rank][1] - self.counts[self.rank][0]
        nmax = max(b - a for a, b in self.counts)
        self.nchunks = (nmax + chunk - 1) // chunk
        self.device = torch.device(device)
        self.frame_shape = tuple(frame_shape)
        # every rank's share fills its chunks exactly: rank r's chunk c is frames [r * share + c * chunk, ...) of the video
        self.exact = all(b - a == self.nchunks * chunk for a, b in self.counts)
        self.buf = None
        if self.on and self.rank == dst:
            self.buf = torch.empty((self.world, self.nchunks * chunk) + self.frame_shape, dtype=torch.uint8, device=self.device)
        self.pad = torch.zeros((chunk,) + self.frame_shape, dtype=torch.uint8, device=self.device) if self.on else None
        self.works, self.pushed = [], 0

    def reset(self):
        """Start the next pass over the video (every collective of the previous one has been waited for by finish())."""
        if self.works:
            raise RuntimeError("reset() before finish()")
        self.pushed = 0

    def push(self, frames: torch.Tensor):
        """frames: the next <= chunk finished frames of this rank ([n, H, W, 3] uint8 on the collective's device; n may be 0)."""
        c = self.pushed
        self.pushed += 1
        if not self.on:
            return
        if c >= self.nchunks:
            raise RuntimeError("more chunks pushed than the schedule holds")
        pad = frames
        if frames.shape[0] != self.chunk:                     # a short or empty chunk: padded to the collective's size
            pad = self.pad.clone()
            pad[: frames.shape[0]] = frames
        pieces = [self.buf[r, c * self.chunk:(c + 1) * self.chunk] for r in range(self.world)] if self.rank == self.dst else None
        self.works.append((dist.gather(pad.contiguous(), pieces, dst=self.dst_global, group=self.group, async_op=True), pad))

    def finish(self, copy: bool = False):
        """Wait for every chunk (ranks that ran out of frames push empty chunks first). Frames in order on dst, else None.
        copy=False: on the exact path the result aliases the receive buffer and is valid until the next pass's first push()."""
        while self.on and self.pushed < self.nchunks:
            self.push(self.pad[:0])
        for w, _ in self.works:
            w.wait()
        self.works = []
        if not self.on:
            return None
        if self.rank != self.dst:
            return None
        if self.exact:
            v = self.buf.view((self.world * self.nchunks * self.chunk,) + self.frame_shape)
            return v.clone() if copy else v
        return torch.cat([self.buf[r, : b - a] for r, (a, b) in enumerate(self.counts)], 0)
