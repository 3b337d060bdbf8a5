"""bench.py -- frames/sec of the CanonSwap generator hot path at 512x512 output on N MI355X.

One "step" = one pass of the per-frame loop body (F -> W.warp -> T -> R -> W.forward -> G,
src/can_swap_pipeline_e2e.py:242-263, debug decodes off) over a batch of B synthetic frames per GPU, inputs
already resident in HBM.  Frames shard across ranks (weak scaling: B frames per GPU per step); the source
identity is broadcast once over RCCL before the timed region and the uint8 output frames are gathered to
rank 0 inside it.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under
torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1), so the command has the same shape at every N.

--streams S (BASELINE configs[4]): S concurrent videos, stream s on the rank group parallel.stream_groups() gives it
(8 GPUs, 4 streams: GPU pair {2s, 2s+1}), one identity per stream, per-stream gather over a sub-communicator;
reports per-stream and aggregate frames/s.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_GFLOP_PER_FRAME = 2374.3      # 2*MAC of every conv on the primary path, as the reference runs it
PEAK_TFLOPS_F16 = 2500.0           # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
# SURVEY 8d, HBM row of the path: per frame two trilinear feature warps (warping_network.py:46-62) of the fp32 volume (8.39 MB in + 8.39 MB
# out each; + one 4.19 MB fp16 copy; the sampling grid costs 0 bytes: it is consumed inside the kernel that computes it) and, in the same
# kernel, the softmax over the 22 mask logits that yields the deformation (dense_motion.py:88-94: 22 x 65536 x 4 B = 5.77 MB of logits per call)
WARP_BYTES_PER_FRAME = 2 * (16.78e6 + 5.77e6) + 4.19e6


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _flush_c_stdio():
    """librccl prints a version banner through C stdio; on a redirected stdout it would sit in libc's buffer until exit and land BEHIND
    the JSON line.  Flushing after the communicators exist keeps the JSON line the last line of rank 0's output."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("CANONSWAP_BENCH_BATCH", "64")),
                    help="frames per launch and GPU (default 64, the engine's maximum: +1.5 % frames/s over 32 - fewer, longer launches leave fewer "
                         "idle workgroup slots at the head and tail of each kernel; profiles/r03_v_batch_sweep.txt)")
    ap.add_argument("--frames", type=int, default=0,
                    help="fixed-size job (BASELINE configs[3]: --frames 1200): one step = one pass over a video of this many frames, "
                         "sharded over the ranks in contiguous blocks (strong scaling). Default 0: every rank runs --batch frames "
                         "per step (weak scaling)")
    ap.add_argument("--streams", type=int, default=0,
                    help="BASELINE configs[4]: this many concurrent videos, each with its own identity, placed by "
                         "parallel.stream_groups (4 streams on 8 GPUs: stream s on GPU pair {2s, 2s+1})")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fixed-job", action="store_true", help="skip the fixed-size job reported next to the weak-scaling figure")
    ap.add_argument("--no-single-frame", action="store_true", help="skip the one-frame-per-call measurement (BASELINE configs[1], latency mode) reported next to the headline")
    ap.add_argument("--no-chain", action="store_true", help="skip the whole-device-side-frame measurement (crops -> M -> generator -> SoftErosion -> paste-back) "
                                                             "and the v2i body reported next to the headline")
    ap.add_argument("--chain-size", default="1080x1920", help="HxW of the original frames the chain pastes back into")
    ap.add_argument("--latency-mode", action="store_true",
                    help="BASELINE configs[1] (--batch 1): cross-workgroup split-K for the launches that cannot fill the chip")
    ap.add_argument("--identities", type=int, default=1,
                    help="this many source identities resident at once, frame g of the job using identity g mod n")
    ap.add_argument("--dump-crc", default="", help="the leader of stream s writes the CRC32 of every gathered frame of the last step "
                                                   "to this path (stream 0) / path.s<s> (tests)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 through the multi-GPU code: RCCL communicator on a world of one, identity broadcast, chunked asynchronous "
                         "device gather of the uint8 frames inside the timed region, sub-communicators for --streams (VERDICT r4 item 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: dry run of the multi-rank flow with all ranks sharing GPU 0 and host-side collectives (test only)")
    return ap.parse_args()


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # same command shape as N = 1: spawn one rank per GPU under torch.distributed.run and hand its exit code back
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import numpy as np
    import torch
    import torch.distributed as dist
    from canonswap_amd import parallel, synth
    from canonswap_amd.can_swap_e2e import can_swapper

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    share_gpu = a.backend == "gloo"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if share_gpu else dev        # where collectives run
    force = bool(a.force_dist) and world == 1               # one rank, every collective still issued
    distd = world > 1 or force
    if distd:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm
    ranks_seen = dist.get_world_size() if distd else 1
    devices = [f"rank {rank}: cuda:{local_rank} {torch.cuda.get_device_name(dev)}"]
    if distd:
        got = [None] * world
        dist.all_gather_object(got, devices[0])
        devices = got

    B, K, Wm = a.batch, a.steps, a.warmup
    S = max(1, a.streams)
    groups = parallel.stream_groups(world, S)               # rank list per stream
    mine = parallel.streams_of_rank(groups, rank)           # [(stream, position in its group, identity slot here)]
    k = len(mine)
    if k < 1 or B % k:
        raise SystemExit(f"--batch {B} must be a multiple of the {k} streams a rank hosts")
    if not distd or (S == 1 and not force):
        comms = [None] * S                                   # S == 1: the default group
    else:
        comms = parallel.make_stream_comms(groups, min_ranks=1 if force else 2)
        for c in comms:                                      # a sub-communicator that cannot form fails here, not in the timed region
            if c is not None:
                dist.barrier(group=c)
    # random-init weights of the real architecture.  The load-time transform (synthesis + pack.build_blobs: tens of seconds of host work,
    # 0.5 GB of packed blobs) runs on rank 0 only; the other ranks of the node read its result from /dev/shm.
    from canonswap_amd import pack
    sds, blobs = None, None
    # The packer is the first local rank of each node and the file name carries the node's name: /dev/shm is node-local.
    shm = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp",
                       f"canonswap_blobs_{socket.gethostname()}_{os.environ.get('MASTER_PORT', os.getpid())}.npz")
    packer = (rank == 0) if share_gpu else (local_rank == 0)
    if packer:
        # + the motion extractor M: the chain leg below drives the generator from M's key-points; the other modules' weights
        # do not depend on its presence (one RNG stream per tensor name)
        sds = synth.to_torch(synth.make_state_dicts(0, modules=synth.MODULES + ("motion_extractor",)))
        blobs = pack.build_blobs(sds)
        if world > 1:
            np.savez(shm, **blobs)
    if world > 1:
        dist.barrier()
        if not packer:
            with np.load(shm) as z:
                blobs = {k: z[k] for k in z.files}
        dist.barrier()
        if packer:
            os.remove(shm)
    sw = can_swapper(type("Cfg", (), {"device_id": local_rank, "flag_force_cpu": False})(), packed_blobs=blobs, max_batch=B,
                     latency_mode=a.latency_mode)
    # BASELINE configs[1] in the same line: a second engine of one frame in latency mode (N = 1 default run only; 0.5 GB of workspace)
    single_on = (rank == 0 and world == 1 and not a.no_single_frame and not a.latency_mode and B > 1 and not a.streams and a.frames <= 0
                 and max(1, min(a.identities, 8)) == 1)
    chain_on = (rank == 0 and world == 1 and not a.no_chain and not a.latency_mode and not a.streams and a.frames <= 0
                and max(1, min(a.identities, 8)) == 1)
    if not single_on:
        del blobs                                            # the one-frame engine is built AFTER the headline timing
    eng = sw.engine

    # one-time broadcast of the source identities (2 KB each); every rank derives T's modulated weights locally
    nid = S if a.streams else max(1, min(a.identities, 8))
    sid = torch.from_numpy(synth.make_identity(7, n=nid)).to(cdev) if rank == 0 else torch.zeros(nid, 512, device=cdev)
    parallel.broadcast_identity(sid, src=0, single_rank=force)
    sid = sid.to(dev)
    if a.streams:
        for s, _, slot in mine:                              # one identity slot per hosted stream
            eng.set_identity(sid[s:s + 1], slot=slot)
    else:
        for j in range(nid):
            eng.set_identity(sid[j:j + 1], slot=j)

    # synthetic inputs resident in HBM: a pool of 4 x B distinct frames.  Frame f of stream s reads pool frame (f + 17 s + 131 step) mod 4B
    # at every step, so the result of a frame does not depend on how many ranks share the job.
    P = 4 * B
    inp = synth.make_frame_inputs(P, seed=1000, size=256)
    pool = {key: torch.from_numpy(inp[key]).to(dev) for key in ("img", "x_t", "x_can")}

    class Plan:
        """Frames of one step on this rank: for every local frame its stream, its index inside the stream's video and its identity slot."""
        def __init__(self, per_stream):
            self.per_stream = per_stream                     # frames per stream and step
            spans = []
            for s, q, slot in mine:
                f0, f1 = parallel.shard_range(per_stream, q, len(groups[s]))
                spans.append((s, slot, f0, f1))
            self.span0 = spans[0][2:]
            n_each = [f1 - f0 for _, _, f0, f1 in spans]
            self.n_local = sum(n_each)
            st, fr, sl = [], [], []
            for j in range(max(n_each) if n_each else 0):    # hosted streams interleaved frame by frame
                for (s, slot, f0, f1) in spans:
                    if f0 + j < f1:
                        st.append(s); fr.append(f0 + j)
                        sl.append(slot if a.streams else (f0 + j) % nid)
            self.slots = sl
            self.base = (torch.tensor(fr, dtype=torch.long) + 17 * torch.tensor(st, dtype=torch.long)).to(dev)
            # launches of this rank: the fewest of at most B frames, all of ONE size that every rank of the stream shares (150 frames per rank
            # at B = 64: 3 x 50, not 64 + 64 + 22); the gather moves chunks of that size
            share = max(parallel.shard_range(per_stream, q, len(groups[s]))[1] - parallel.shard_range(per_stream, q, len(groups[s]))[0]
                        for s, _, _ in mine for q in range(len(groups[s])))
            self.chunk = parallel.equal_chunks(share * k, B)[0]
            self.chunks = [(t0, min(self.chunk, self.n_local - t0)) for t0 in range(0, self.n_local, self.chunk)]
            self.gather = None                               # ChunkedFrameGather of this plan: built once, outside any timed region

    weak = a.frames <= 0
    plan = Plan((B // k) * len(groups[0]) if weak else a.frames)
    fixed_T = 0 if (a.frames > 0 or a.streams or a.no_fixed_job) else (1200 if world > 1 else 300)      # configs[3] / configs[2]
    fplan = Plan(fixed_T) if fixed_T else None
    out_u8 = torch.empty(max(plan.n_local, fplan.n_local if fplan else 0, 1), 512, 512, 3, dtype=torch.uint8, device=dev)
    gather_on = distd and k == 1 and (len(groups[mine[0][0]]) > 1 or force)
    my_comm = comms[mine[0][0]]
    if gather_on:                                            # receive buffers (943 MB on the leader for 1200 frames) allocated once, here
        for pl in (plan, fplan):
            if pl is not None:
                pl.gather = parallel.ChunkedFrameGather(pl.per_stream, pl.chunk, device=cdev, group=my_comm, single_rank=force)

    def step(pl, i, gather=None, out_f32=None):
        """One step: this rank's frames in chunks of B; finished chunks go to the stream's leader asynchronously."""
        for t0, n in pl.chunks:
            idx = (pl.base[t0:t0 + n] + 131 * i) % P
            eng.swap_frames(pool["img"][idx], pool["x_t"][idx], pool["x_can"][idx], None, want_f32=False, want_u8=True,
                            out_u8=out_u8[t0:t0 + n], out_f32=out_f32 if t0 == 0 else None, slots=pl.slots[t0:t0 + n])
            out_f32 = None
            if gather is not None:
                gather.push(out_u8[t0:t0 + n].to(cdev))
        return gather.finish() if gather is not None else None

    def sync():
        torch.cuda.synchronize(dev)
        if distd:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(pl, steps):
        """EXACTLY `steps` steps between barrier + synchronize on both sides; returns (max-over-ranks seconds, per-rank seconds, frames on
        the stream leader)."""
        sync()
        t0 = time.perf_counter()
        gathered = None
        for i in range(steps):
            g = pl.gather if gather_on else None
            if g is not None:
                g.reset()
            r = step(pl, i, g)
            gathered = r if gather_on else out_u8[:pl.n_local]
        sync()
        dt = time.perf_counter() - t0
        per_rank = [dt]
        if distd:
            t = torch.tensor([dt], dtype=torch.float64, device=cdev)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            per_rank = [float(x.item()) for x in allt]
        return max(per_rank), per_rank, gathered

    for i in range(Wm):
        step(plan, i)
    dt, per_rank, gathered = timed(plan, K)
    _flush_c_stdio()
    if a.dump_crc:                                            # tests: CRC32 of every frame of the last step, per stream, on its leader
        import zlib
        for h, (s, q, _) in enumerate(mine):
            if k == 1:
                fr = gathered if q == 0 else None            # the group's first rank holds the gathered stream
            else:
                fr = out_u8[h:plan.n_local:k]                # hosted streams are interleaved frame by frame
            if fr is None:
                continue
            fr = fr.cpu().numpy()
            assert fr.shape[0] == plan.per_stream, (fr.shape, plan.per_stream)
            crcs = [zlib.crc32(fr[j].tobytes()) for j in range(fr.shape[0])]
            for path in ([a.dump_crc] if s == 0 else []) + [f"{a.dump_crc}.s{s}"]:
                with open(path, "w") as f:
                    json.dump(crcs, f)

    fixed = None
    if fplan is not None:
        step(fplan, 0)                                       # warm-up pass (allocator, ragged last chunk)
        fdt, fper, fg = timed(fplan, 1)
        if rank == 0:
            assert fg.shape[0] == fixed_T
            fixed = {"workload": f"BASELINE configs[{3 if world > 1 else 2}]: one 512x512 video of {fixed_T} frames, " +
                                 (f"sharded over {world} GPUs in contiguous blocks, uint8 frames gathered to rank 0" if world > 1 else "batched on 1 GPU"),
                     "frames": fixed_T, "seconds": round(fdt, 4), "value": round(fixed_T / fdt, 3), "unit": "frames/s", "scaling": "strong",
                     "frames_per_launch": fplan.chunk, "launches_per_rank": len(fplan.chunks), "per_rank_seconds": [round(x, 4) for x in fper],
                     # efficiency against the same kind of job on one GPU = value_per_gpu / (the N = 1 line's fixed_job.value)
                     "value_per_gpu": round(fixed_T / fdt / world, 3)}

    # ---- BASELINE configs[1]: one 512x512 frame per call, latency mode (cs_set_latency_mode: DESIGN 5.8), inputs resident, every call synchronous
    # with the next only through the stream (the caller hands over frame after frame); 10 untimed + 200 timed calls
    single, single_out = None, None
    if single_on:
        sw1 = can_swapper(type("Cfg", (), {"device_id": local_rank, "flag_force_cpu": False})(), packed_blobs=blobs, max_batch=1, latency_mode=True)
        del blobs
        e1 = sw1.engine
        e1.set_identity(sid[0:1], slot=0)
        o1 = torch.empty(1, 512, 512, 3, dtype=torch.uint8, device=dev)
        f1 = torch.empty(1, 3, 512, 512, dtype=torch.float32, device=dev)

        def one(j, want_f32=False):
            e1.swap_frames(pool["img"][j:j + 1], pool["x_t"][j:j + 1], pool["x_can"][j:j + 1], None, want_f32=want_f32, want_u8=True, out_u8=o1,
                           out_f32=f1 if want_f32 else None, slots=[0])
        for i in range(10):
            one(i % P)
        torch.cuda.synchronize(dev)
        n1, t1 = 200, time.perf_counter()
        for i in range(n1):
            one((131 * i) % P)
        torch.cuda.synchronize(dev)
        d1 = time.perf_counter() - t1
        one(0, want_f32=True)                                # pool frame 0 for the parity leg below
        torch.cuda.synchronize(dev)
        single_out = f1.cpu()
        single = {"workload": "BASELINE configs[1]: one 512x512 frame per call on 1 GPU, latency mode (cs_set_latency_mode), inputs resident in HBM",
                  "frames": n1, "ms_per_frame": round(d1 / n1 * 1e3, 3), "value": round(n1 / d1, 3), "unit": "frames/s",
                  "conv_roofline_frac_end_to_end": round(ALGO_GFLOP_PER_FRAME * 1e9 * (n1 / d1) / (PEAK_TFLOPS_F16 * 1e12), 4)}
        e1.close()


    # ---- the whole device-side frame (SURVEY 8f rows N1-N3 around the generator): uint8 512x512 crops resident in HBM ->
    # cs_prepare_crops -> cs_motion_extract + cs_motion_keypoints -> cs_swap_frames_ids -> cs_soft_erosion_frames -> cs_paste_back_batch -> uint8
    # frames of the original size, B frames per launch; next to it the generator alone on the SAME key-points, and the v2i body (row N4)
    chain, animate = None, None
    if chain_on:
        import csv as _csv
        import tempfile
        from canonswap_amd.chain import FrameChain
        Ho, Wo = (int(v) for v in a.chain_size.lower().split("x"))
        fc = FrameChain(sw)
        smooth = synth.make_smooth_images(B, seed=2100, size=512)                        # crops M can tell apart (white noise pools to one feature)
        crops = torch.from_numpy(np.ascontiguousarray((smooth.transpose(0, 2, 3, 1) * 255).astype(np.uint8))).to(dev)
        del smooth
        yy, xx = np.mgrid[0:512, 0:512].astype(np.float32)
        rr = np.random.Generator(np.random.PCG64(77))
        mk = np.stack([(((xx - rr.uniform(216, 296)) / rr.uniform(120, 190)) ** 2 + ((yy - rr.uniform(216, 296)) / rr.uniform(150, 215)) ** 2 <= 1)
                       for _ in range(B)]).astype(np.uint8)                              # face-parsing labels: one 0/1 ellipse per frame
        masks = torch.from_numpy(mk).to(dev)
        ori = torch.randint(0, 256, (B, Ho, Wo, 3), dtype=torch.uint8, device=dev)
        outf = torch.empty_like(ori)
        Ms = np.zeros((B, 2, 3))
        for j in range(B):                                                               # crop -> frame: a face 0.35-0.5 of the frame height
            sc, th = rr.uniform(0.35, 0.5) * Ho / 512.0, rr.uniform(-0.2, 0.2)
            Ms[j] = [[sc * np.cos(th), -sc * np.sin(th), rr.uniform(0.25, 0.45) * Wo], [sc * np.sin(th), sc * np.cos(th), rr.uniform(0.1, 0.3) * Ho]]
        slots0 = [0] * B

        def chain_step():
            return fc(crops, masks, Ms, ori, slots=slots0, out=outf, keep=True)
        for _ in range(2):
            r0 = chain_step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(K):
            chain_step()
        torch.cuda.synchronize(dev)
        cdt = time.perf_counter() - t0
        # ... and with stage A (staging, M, key-points, soft masks) of batch k + 1 on a side stream beside the generator of batch k - the
        # reference runs that stage as a pre-pass over the whole video (can_swap_pipeline_e2e.py:196-197), so nothing of frame k + 1 depends on frame k
        crops_b = crops.clone()                              # the "next" batch is another tensor object: FrameChain matches a prefetch by identity

        def pipe_run(n):
            cur, nxt = crops, crops_b
            fc.prefetch(cur, masks)
            for _ in range(n):
                fc.prefetch(nxt, masks)                      # batch k + 1 is queued on the side stream before the launches of batch k
                fc(cur, masks, Ms, ori, slots=slots0, out=outf)
                cur, nxt = nxt, cur
            fc.drop_prefetches()
        pipe_run(2)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        pipe_run(K)
        torch.cuda.synchronize(dev)
        pdt = time.perf_counter() - t0
        del crops_b
        I_c, xt_c, xc_c = r0["I"], r0["x_t"].clone(), r0["x_can"].clone()
        gen_u8 = torch.empty(B, 512, 512, 3, dtype=torch.uint8, device=dev)

        def gen_step():
            eng.swap_frames(I_c, xt_c, xc_c, None, want_f32=False, want_u8=True, out_u8=gen_u8, slots=slots0)
        for _ in range(2):
            gen_step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(K):
            gen_step()
        torch.cuda.synchronize(dev)
        gdt = time.perf_counter() - t0
        # per-stage device time of one chain step: HIP events around every launch, summed by stage
        tmp = tempfile.NamedTemporaryFile(suffix=".csv", delete=False); tmp.close()
        old_csv = os.environ.get("CANONSWAP_PROFILE_CSV")
        os.environ["CANONSWAP_PROFILE_CSV"] = tmp.name
        eng.profile_begin()
        chain_step()
        eng.profile_end()
        if old_csv is None:
            del os.environ["CANONSWAP_PROFILE_CSV"]
        else:
            os.environ["CANONSWAP_PROFILE_CSV"] = old_csv
        stage_ms = {"prepare_crops": 0.0, "motion_extractor": 0.0, "m_keypoints": 0.0, "generator": 0.0, "soft_erosion": 0.0, "paste_back": 0.0}
        with open(tmp.name) as f:
            for row in _csv.DictReader(f):
                lab, ms = row["label"], float(row["ms"])
                key = ("prepare_crops" if lab == "prepare_crops" else "m_keypoints" if lab == "m_keypoints" else
                       "motion_extractor" if (lab.startswith("M.") or lab.startswith("m_")) else "soft_erosion" if lab == "soft_erosion" else
                       "paste_back" if lab == "paste_back_batch" else "generator")
                stage_ms[key] += ms
        os.remove(tmp.name)
        px = Ho * Wo
        stage_bytes = {"prepare_crops": 2 * 786432.0, "soft_erosion": 262144.0 + 4 * 262144.0, "paste_back": 2.0 * px * 3 + 786432 + 4 * 262144}
        chain = {"workload": f"whole device-side frame, {B} frames per launch: uint8 512x512 crops in HBM -> INTER_AREA 256x256 + /255 -> motion extractor M "
                             f"+ transform_keypoint -> F->W->T->R->W->G -> SoftErosion(21, 0.9, 3) of the 0/1 face mask -> warpAffine paste-back into {Ho}x{Wo} "
                             "uint8 frames (can_swap_pipeline_e2e.py:111-125, 196, 242-283 without its host round trips)",
                 "frames": K * B, "value": round(K * B / cdt, 3), "unit": "frames/s", "ms_per_step": round(cdt / K * 1e3, 3),
                 "generator_alone_same_keypoints": round(K * B / gdt, 3), "ratio_to_generator": round(gdt / cdt, 4),
                 "value_overlapped": round(K * B / pdt, 3), "ratio_to_generator_overlapped": round(gdt / pdt, 4),
                 "overlapped": "staging + M + key-points + soft masks of batch k + 1 on a side HIP stream beside the generator of batch k (FrameChain.prefetch)",
                 "stage_ms_per_step": {k: round(v, 3) for k, v in stage_ms.items()},
                 "stage_rates": {"prepare_crops_GBps": round(stage_bytes["prepare_crops"] * B / max(stage_ms["prepare_crops"], 1e-9) / 1e6, 1),
                                 "soft_erosion_GBps": round(stage_bytes["soft_erosion"] * B / max(stage_ms["soft_erosion"], 1e-9) / 1e6, 1),
                                 "soft_erosion_fp32_TFLOPs": round(2 * 441 * 262144 * 3 * B / max(stage_ms["soft_erosion"], 1e-9) / 1e9, 2),
                                 "paste_back_GBps": round(stage_bytes["paste_back"] * B / max(stage_ms["paste_back"], 1e-9) / 1e6, 1),
                                 "motion_extractor_algorithmic_TFLOPs": round(11.6 * B / max(stage_ms["motion_extractor"], 1e-9), 1)},
                 "stage_rates_basis": "algorithmic bytes per frame: prepare_crops 786 KB in + 786 KB out; soft_erosion 262 KB of labels in + 1 MB soft mask out, "
                                      "0.69 GFLOP of fp32 FMAs (three 21x21 cone convolutions, crop.py:29-41); paste_back original frame in + out, crop and "
                                      "mask in; M 11.6 GFLOP (SURVEY 8f), run in split precision (three fp16 MFMA passes)"}
        del ori, outf, crops, masks, fc
        # ---- v2i body (row N4; can_swap_pipeline_v2i.py:311-312): warp_decode of ONE swapped canonical volume under B driving key-point sets
        with torch.no_grad():
            f_can = eng.extract_feature_3d(pool["img"][0:1])
        xs = pool["x_can"][0:1].contiguous()
        kd = pool["x_t"][:B].contiguous()

        def v2i_step():
            eng.animate_frames(f_can, xs, kd, want_f32=False, want_u8=True, out_u8=gen_u8)
        for _ in range(2):
            v2i_step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(K):
            v2i_step()
        torch.cuda.synchronize(dev)
        vdt = time.perf_counter() - t0
        V2I_GFLOP = 342.07 + 891.57                        # W.forward + G per frame
        animate = {"workload": f"v2i per-frame body (can_swap_pipeline_v2i.py:311-312): warp_decode of one swapped canonical volume under {B} driving "
                               "key-point sets per launch (cs_animate_frames, shared volume)",
                   "frames": K * B, "value": round(K * B / vdt, 3), "unit": "frames/s", "ms_per_step": round(vdt / K * 1e3, 3),
                   "conv_roofline_frac_end_to_end": round(V2I_GFLOP * 1e9 * (K * B / vdt) / (PEAK_TFLOPS_F16 * 1e12), 4)}
        del gen_u8

    # ---- roofline of the dominant kernel family (conv_halo_kernel): HIP events around every launch, same workload
    prof, sparse_ms, sparse_n = None, 0.0, 0
    if rank == 0:
        import csv as _csv2
        import tempfile as _tf
        ptmp = _tf.NamedTemporaryFile(suffix=".csv", delete=False); ptmp.close()
        user_csv = os.environ.get("CANONSWAP_PROFILE_CSV")
        if user_csv is None:                                 # per-launch records of this pass: the sparse-motion sampler's own row (below)
            os.environ["CANONSWAP_PROFILE_CSV"] = ptmp.name
        eng.profile_begin()
        for i in range(K):
            step(plan, i)
        prof = eng.profile_end()
        if user_csv is None:
            del os.environ["CANONSWAP_PROFILE_CSV"]
        with open(user_csv or ptmp.name) as f:
            for row in _csv2.DictReader(f):
                if row["label"] == "dm_sparse":
                    sparse_ms += float(row["ms"]); sparse_n += 1
        os.remove(ptmp.name)
    if distd:
        dist.barrier()

    cpu, parity = None, None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import canonswap_ref as O          # cpu_baseline leg: the oracle timed on this node's host cores
        cores = min(os.cpu_count() or 1, 16)           # threads used ("cores"): the fastest count on the GPU box's host (profiles/r05_r_cpu_threads.txt: 0.59 frames/s at 16, 0.45 at 32, 0.14 at 128 of 256 cores)
        torch.set_num_threads(cores)
        # parity of THIS binary in THIS run: first and last frame of the first launch (step 0) against the fp32 oracle
        from canonswap_amd import pack
        t0, n = plan.chunks[0]
        o32 = torch.empty(n, 3, 512, 512, dtype=torch.float32, device=dev)
        idx = (plan.base[:n] % P).clone()                  # step 0, first launch
        # ... plus the frame of the WHOLE input pool with the lowest PSNR in the survey of all its 4 x B frames (tests/diag/psnr_pool.py,
        # committed as profiles/psnr_worst_frame.json: the pool and the identity are fixed by their seeds).  It rides in position 1 of this
        # parity launch when the first launch does not hold it (a frame's bits do not depend on the launch it rides in).
        sample, wpath = {0, n - 1}, os.path.join(ROOT, "profiles", "psnr_worst_frame.json")
        worst_known, worst_pos = None, None
        if os.path.exists(wpath) and not a.streams and nid == 1:
            wj = json.load(open(wpath))
            if int(wj.get("batch", -1)) == n and n >= 3:
                worst_known = int(wj["worst_frame"])
                hit = (idx == worst_known).nonzero()
                if hit.numel():
                    worst_pos = int(hit[0])
                else:
                    worst_pos = 1
                    idx[1] = worst_known
                sample.add(worst_pos)
        eng.swap_frames(pool["img"][idx], pool["x_t"][idx], pool["x_can"][idx], None, want_f32=True, want_u8=True,
                        out_u8=out_u8[:n], out_f32=o32, slots=plan.slots[:n])
        torch.cuda.synchronize(dev)
        idx, ids_cpu = idx.cpu(), sid.cpu()
        osd = sds
        worst, mad, cargs, cid = 1e9, 0.0, None, None
        for j in sorted(sample):
            cargs = [torch.from_numpy(inp[key][int(idx[j]):int(idx[j]) + 1]) for key in ("img", "x_t", "x_can")]
            row = mine[j % k][0] if a.streams else plan.slots[j]
            cid = ids_cpu[row:row + 1]
            with torch.no_grad():
                ref = O.swap_frame(osd, *cargs, cid)["out"]
            worst = min(worst, O.psnr(o32[j:j + 1].cpu(), ref))
            d = out_u8[j:j + 1].cpu().numpy().astype(np.float64) - O.parse_output(ref).astype(np.float64)
            mad = max(mad, float(np.abs(d).mean()))
        if single is not None:                              # the one-frame engine's pool frame 0 against the same oracle
            with torch.no_grad():
                ref0 = O.swap_frame(osd, *[torch.from_numpy(inp[key][0:1]) for key in ("img", "x_t", "x_can")], ids_cpu[0:1])["out"]
            single["psnr_db_pool_frame_0"] = round(O.psnr(single_out, ref0), 2)
        parity = {"psnr_db_min": round(worst, 2), "u8_mean_abs_diff": round(mad, 4),
                  "parity_sample": f"pool frames {[int(idx[j]) for j in sorted(sample)]} in one {n}-frame launch vs the fp32 CPU oracle" +
                                   (f" (pool frame {worst_known}: the worst of the {wj.get('pool_frames', n)} surveyed in profiles/psnr_worst_frame.json)"
                                    if worst_known is not None else "")}
        n_cpu, t1 = 8, time.perf_counter()             # about 11 s of CPU work (the two parity frames above were the warm-up)
        for _ in range(n_cpu):
            O.swap_frame(osd, *cargs, cid)
        cdt = time.perf_counter() - t1
        cpu = {"value": round(n_cpu / cdt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
               "sample": f"{n_cpu} frames (2 warm-up), batch 1, fp32 PyTorch-CPU restatement of the same path (oracle/)"}

    traffic, traffic_src, warp_traffic = None, None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if rank == 0 and os.path.exists(tpath):      # PMC counters cannot be read from inside this process: committed rocprofv3 passes
        t = json.load(open(tpath))
        if int(t.get("batch", -1)) == B:
            traffic = t["fetch_bytes_per_conv_launch_x2"] + t["write_bytes_per_conv_launch_raw"]
            traffic_src = "profiles/hbm_traffic.json: " + t["source"]
            if "warp_fetch_bytes_per_launch_x2" in t:
                warp_traffic = t["warp_fetch_bytes_per_launch_x2"] + t["warp_write_bytes_per_launch_raw"]

    if rank == 0:
        frames = K * plan.per_stream * S
        fps = frames / dt
        conv_s = prof["conv_ms"] / 1e3
        achieved = prof["conv_flops"] / conv_s / 1e12
        if a.streams:
            wl = (f"BASELINE configs[4]: {S} concurrent 512x512 video streams, stream s on ranks {groups[0] if S == 1 else '/'.join(str(g) for g in groups)}, "
                  "one identity per stream")
        elif weak:
            wl = (f"BASELINE configs[2]: 512x512 video, {B} frames batched per launch on each GPU, steady state over {frames} frames "
                  f"({K} launches per GPU)")
        else:
            wl = f"BASELINE configs[3]: 512x512 video of {plan.per_stream} frames sharded over {world} GPU(s) in contiguous blocks"
        line = {
            "metric": "frames/sec at 512x512 (generator hot path F->W->T->R->W->G)",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "ranks_seen": ranks_seen, "backend": (a.backend if distd else None), "collectives": ("broadcast + chunked async gather" + (
                " on a one-rank communicator (--force-dist)" if force else "")) if distd else None, "devices": devices, "per_rank_seconds": [round(x, 4) for x in per_rank],
            "config": {"workload": wl + " (256x256 crops in, random-init weights of the real architecture)",
                       "frames_per_step": plan.per_stream * S, "frames_per_launch_per_gpu": B, "frames_total": frames,
                       "parallelism": f"frame-shard x{world}" + (f", {S} streams" if a.streams else ""), "identities_resident": nid,
                       "latency_mode": bool(a.latency_mode), "accumulate": "fp32", "debug_decodes": False},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_TFLOPS_F16, 4), "traffic": traffic, "traffic_unit": "HBM bytes per conv launch",
                         "traffic_source": traffic_src,
                         "kernel": "conv_wide_kernel / conv_halo_kernel / vol32_kernel (every convolution launch of the step)", "launches_per_step": prof["conv_launches"] // K,
                         "avg_launch_us": round(prof["conv_ms"] * 1e3 / prof["conv_launches"], 2),
                         "algorithmic_gflop_per_frame": round(prof["conv_flops"] / (K * plan.n_local) / 1e9, 1),
                         # MFMA work actually issued (padded channel counts, phase-decomposed up-sampling convs): the utilisation of the
                         # matrix pipe, as opposed to the algorithmic rate above
                         "executed_tflops": round(prof["exec_flops"] / conv_s / 1e12, 2),
                         "executed_frac": round(prof["exec_flops"] / conv_s / 1e12 / PEAK_TFLOPS_F16, 4),
                         "other_kernels_ms_per_step": round((prof["other_ms"] + prof["warp_ms"]) / K, 3),
                         "conv_ms_per_step": round(prof["conv_ms"] / K, 3),
                         "end_to_end_frac": round(ALGO_GFLOP_PER_FRAME * 1e9 * fps / world / (PEAK_TFLOPS_F16 * 1e12), 4)},
            # the HBM-bound row of the path: softmax -> deformation -> trilinear feature warp in one kernel (dm_softmax_warp_kernel)
            "warp_roofline": {"bound": "hbm", "kernel": "dm_softmax_warp_kernel (mask softmax + deformation + feature warp; the logits arrive as 10 x 22 "
                                                        "fp32 partial sums per four voxels from the mask conv's in-tile kw sum: 2.5x the algorithmic 5.77 MB)",
                              "algorithmic_mb_per_frame": round(WARP_BYTES_PER_FRAME / 1e6, 2),
                              "achieved": round(WARP_BYTES_PER_FRAME * plan.n_local * K / (prof["warp_ms"] / 1e3) / 1e9, 1),
                              "peak": 8000.0, "unit": "GB/s",
                              "frac": round(WARP_BYTES_PER_FRAME * plan.n_local * K / (prof["warp_ms"] / 1e3) / 8e12, 4),
                              "avg_launch_us": round(prof["warp_ms"] * 1e3 / max(prof["warp_launches"], 1), 2), "traffic": warp_traffic,
                              "traffic_unit": "HBM bytes per warp launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/hbm_traffic.json)"},
            # the other HBM row SURVEY 8d names: the sparse-motion sampler (dense_motion.py:29-65: 22 deformed copies of the 4-channel compressed
            # volume + 22 heat maps, concatenated) - per call 4 x 65536 x 2 B read + 22 x 4 x 65536 x 2 B written (the 17 MB grid is generated from
            # the 21 key-points, never read) = 12.06 MB at 2-byte storage; two calls per frame
            "sparse_roofline": {"bound": "hbm", "kernel": "dm_sparse_kernel (sparse motions + deformed features + heat maps + concat into the hourglass input)",
                                "algorithmic_mb_per_call": 12.06, "calls_per_frame": 2,
                                "achieved": round(12.06e6 * 2 * K * plan.n_local / max(sparse_ms, 1e-9) / 1e6, 1) if sparse_n else None,
                                "peak": 8000.0, "unit": "GB/s",
                                "frac": round(12.06e6 * 2 * K * plan.n_local / max(sparse_ms, 1e-9) / 1e6 / 8000.0, 4) if sparse_n else None,
                                "avg_launch_us": round(sparse_ms * 1e3 / sparse_n, 2) if sparse_n else None},
            "cpu_baseline": cpu,
        }
        if parity:
            line.update(parity)
        if fixed:
            line["fixed_job"] = fixed
        if single:
            line["single_frame"] = single
        if chain:
            line["chain"] = chain
        if animate:
            line["v2i_body"] = animate
        if a.streams:
            per = []
            for s, g in enumerate(groups):
                ts = max(per_rank[r] for r in g)
                per.append({"stream": s, "ranks": g, "frames": K * plan.per_stream, "value": round(K * plan.per_stream / ts, 3), "unit": "frames/s"})
            line["streams"] = per
        _flush_c_stdio()
        print(json.dumps(line), flush=True)
    if distd:
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0 and parity and not parity["psnr_db_min"] >= 50.0:
        raise SystemExit(f"bench: parity below the 50 dB gate (psnr_db_min {parity['psnr_db_min']})")


if __name__ == "__main__":
    main()
