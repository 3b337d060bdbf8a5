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


def test_stream_groups_placement():
    """stream s -> GPU pair {2s, 2s+1} (8 GPUs, 4 streams); fewer ranks than streams: whole streams per rank."""
    assert parallel.stream_groups(8, 4) == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert parallel.stream_groups(4, 2) == [[0, 1], [2, 3]]
    assert parallel.stream_groups(8, 1) == [list(range(8))]
    assert parallel.stream_groups(2, 4) == [[0], [1], [0], [1]]
    assert parallel.streams_of_rank(parallel.stream_groups(8, 4), 5) == [(2, 1, 0)]
    assert parallel.streams_of_rank(parallel.stream_groups(2, 4), 1) == [(1, 0, 0), (3, 0, 1)]
    for bad in ((6, 4), (3, 2), (2, 3)):
        try:
            parallel.stream_groups(*bad)
        except ValueError:
            continue
        raise AssertionError(bad)


def _stream_worker(rank, world, port, n_streams, per_stream, chunk, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        groups = parallel.stream_groups(world, n_streams)
        comms = parallel.make_stream_comms(groups)              # collective: every rank creates every sub-communicator
        ids = torch.stack([torch.full((512,), float(s + 1)) for s in range(n_streams)]) if rank == 0 else torch.zeros(n_streams, 512)
        parallel.broadcast_identity(ids, src=0)                 # one broadcast of every stream's identity over the default group
        (s, pos, slot), = parallel.streams_of_rank(groups, rank)
        assert slot == 0
        tag = int(ids[s, 0].item())                             # stand-in for the engine: frame value = 10 * frame index + identity tag
        a, b = parallel.shard_range(per_stream, pos, len(groups[s]))
        g = parallel.ChunkedFrameGather(per_stream, chunk, frame_shape=(2, 2, 3), group=comms[s], dst=0)
        for t0 in range(a, b, chunk):
            n = min(chunk, b - t0)
            g.push(torch.stack([torch.full((2, 2, 3), (10 * i + tag) % 256, dtype=torch.uint8) for i in range(t0, t0 + n)]))
        out = g.finish()
        if pos == 0:
            q.put((s, rank, out[:, 0, 0, 0].tolist()))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_two_streams_on_four_ranks_gather_per_stream():
    """configs[4] shape at world 4: stream s on ranks {2s, 2s+1}; each stream's frames arrive on ITS leader (ranks 0 and 2), in frame
    order, carrying that stream's identity; the two gathers run on separate sub-communicators."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, n_streams, per_stream, port = 4, 2, 7, _free_port()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, n_streams, per_stream, 2, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(n_streams))
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert [(s, r) for s, r, _ in got] == [(0, 0), (1, 2)]
    for s, _, frames in got:
        assert frames == [(10 * i + s + 1) % 256 for i in range(per_stream)]


def test_equal_chunks():
    """A rank's share runs as the fewest launches of at most the engine's batch, all of one size."""
    assert parallel.equal_chunks(150, 64) == (50, [(0, 50), (50, 50), (100, 50)])
    assert parallel.equal_chunks(64, 64) == (64, [(0, 64)])
    assert parallel.equal_chunks(300, 64) == (60, [(t, 60) for t in range(0, 300, 60)])
    size, chunks = parallel.equal_chunks(151, 64)
    assert size == 51 and sum(n for _, n in chunks) == 151 and max(n for _, n in chunks) <= 64
    assert parallel.equal_chunks(0, 64)[1] == []


def _job_worker(rank, world, port, n_frames, max_batch, passes, q):
    """BASELINE configs[3] shape (1200 frames over 8 ranks) with tiny frames: shard_range -> equal chunks -> ONE ChunkedFrameGather reused
    over several passes; the leader gets every frame in order without a copy."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = parallel.shard_range(n_frames, rank, world)
        share = max(parallel.shard_range(n_frames, r, world)[1] - parallel.shard_range(n_frames, r, world)[0] for r in range(world))
        size, _ = parallel.equal_chunks(share, max_batch)
        chunks = [(t0, min(size, (b - a) - t0)) for t0 in range(0, b - a, size)]
        g = parallel.ChunkedFrameGather(n_frames, size, frame_shape=(1, 2, 3))
        buf0 = g.buf.data_ptr() if rank == 0 else None
        res = []
        for ps in range(passes):
            g.reset()
            for t0, n in chunks:
                idx = torch.arange(a + t0, a + t0 + n)
                fr = ((idx + 7 * ps) % 251).to(torch.uint8).view(-1, 1, 1, 1).expand(n, 1, 2, 3).contiguous()
                g.push(fr)
            out = g.finish()
            if rank == 0:
                assert out.shape[0] == n_frames
                res.append((out[:, 0, 0, 0].tolist(), g.exact, out.data_ptr() == buf0))
            else:
                assert out is None
        if rank == 0:
            q.put((size, len(chunks), res))
    finally:
        dist.destroy_process_group()


def test_1200_frames_on_8_ranks_equal_chunks_reused_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_frames, world, port, passes = 1200, 8, _free_port(), 2
    procs = [ctx.Process(target=_job_worker, args=(r, world, port, n_frames, 64, passes, q)) for r in range(world)]
    for p in procs:
        p.start()
    size, nchunks, res = q.get(timeout=300)
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert (size, nchunks) == (50, 3)                         # 150 frames per rank: 3 launches of 50
    for ps, (frames, exact, is_view) in enumerate(res):
        assert exact and is_view                              # whole chunks on every rank: the receive buffer IS the video, no copy
        assert frames == [(i + 7 * ps) % 251 for i in range(n_frames)]


def test_ragged_job_on_3_ranks_reused_gather():
    """100 frames over 3 ranks (34 + 33 + 33) at a 16-frame maximum: shares that are not whole chunks take the copying path."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_frames, world, port = 100, 3, _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, world, port, n_frames, 16, 2, q)) for r in range(world)]
    for p in procs:
        p.start()
    size, nchunks, res = q.get(timeout=300)
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert size == 12 and nchunks == 3
    for ps, (frames, exact, _) in enumerate(res):
        assert not exact
        assert frames == [(i + 7 * ps) % 251 for i in range(n_frames)]
