"""world_size-2 gloo test of the N>1 path: shard -> broadcast identity -> per-rank work -> gather."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from canonswap_amd import parallel


def test_shard_range_partitions():
    for n in (0, 1, 7, 150, 1200, 1201):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sid = torch.arange(512, dtype=torch.float32) / 512 if rank == 0 else torch.zeros(512)
        parallel.broadcast_identity(sid, src=0)
        a, b = parallel.shard_range(n_frames, rank, world)
        # stand-in for the engine: every "frame" is filled with (frame index + identity checksum) mod 256
        tag = int(sid.sum().item())
        local = torch.stack([torch.full((4, 4, 3), (i + tag) % 256, dtype=torch.uint8) for i in range(a, b)]) if b > a \
            else torch.zeros((0, 4, 4, 3), dtype=torch.uint8)
        out = parallel.gather_frames(local, n_frames, dst=0)
        if rank == 0:
            q.put((tag, out[:, 0, 0, 0].tolist()))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_frames, world, port = 7, 2, _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    tag, got = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert tag == int((torch.arange(512, dtype=torch.float32) / 512).sum().item())
    assert got == [(i + tag) % 256 for i in range(n_frames)]


def _chunk_worker(rank, world, port, n_frames, chunk, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = parallel.shard_range(n_frames, rank, world)
        g = parallel.ChunkedFrameGather(n_frames, chunk, frame_shape=(2, 2, 3))
        for t0 in range(a, b, chunk):        # ranks push different numbers of chunks; finish() pads the schedule
            n = min(chunk, b - t0)
            g.push(torch.stack([torch.full((2, 2, 3), i % 256, dtype=torch.uint8) for i in range(t0, t0 + n)]))
        out = g.finish()
        if rank == 0:
            q.put(out[:, 0, 0, 0].tolist())
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_chunked_gather_world2_ragged():
    """ChunkedFrameGather: 13 frames over 2 ranks (7 + 6) in chunks of 3 -> 3 collectives, the last ones short / empty."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_frames, world, port = 13, 2, _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, world, port, n_frames, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got == list(range(n_frames))


def test_chunked_gather_single_process_is_a_no_op():
    g = parallel.ChunkedFrameGather(5, 2, frame_shape=(2, 2, 3))
    g.push(torch.zeros(2, 2, 2, 3, dtype=torch.uint8))
    assert g.finish() is None
