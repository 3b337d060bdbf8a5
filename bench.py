"""bench.py -- frames/sec of the CanonSwap generator hot path at 512x512 output on N MI355X.

One "step" = one pass of the per-frame loop body (F -> W.warp -> T -> R -> W.forward -> G,
src/can_swap_pipeline_e2e.py:242-263, debug decodes off) over a batch of B synthetic frames per GPU, inputs
already resident in HBM.  Frames shard across ranks (weak scaling: B frames per GPU per step); the source
identity is broadcast once over RCCL before the timed region and the uint8 output frames are gathered to
rank 0 inside it.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_GFLOP_PER_FRAME = 2374.3      # 2*MAC of every conv on the primary path, as the reference runs it
PEAK_TFLOPS_F16 = 2500.0           # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("CANONSWAP_BENCH_BATCH", "32")))
    ap.add_argument("--frames", type=int, default=0,
                    help="fixed-size job (BASELINE configs[3]: --frames 1200): one step = one pass over a video of this many frames, "
                         "sharded over the ranks in contiguous blocks (strong scaling). Default 0: every rank runs --batch frames "
                         "per step (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8-weights", action="store_true",
                    help="BASELINE configs[4] numerics: conv weights quantised to e4m3 with per-out-channel scales (expanded to f16 for "
                         "the MFMA, whose operands must share a format class); activations f16")
    ap.add_argument("--latency-mode", action="store_true",
                    help="BASELINE configs[1] (--batch 1): cross-workgroup split-K for the launches that cannot fill the chip")
    ap.add_argument("--identities", type=int, default=1,
                    help="configs[4]: this many source identities resident at once, frames of a launch cycling through them")
    ap.add_argument("--dump-crc", default="", help="rank 0 writes the CRC32 of every gathered frame of the last step here (tests)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: dry run of the multi-rank flow with all ranks sharing GPU 0 and host-side collectives (test only)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from canonswap_amd import parallel, synth
    from canonswap_amd.can_swap_e2e import can_swapper

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs a {a.gpus}-rank launch (WORLD_SIZE={world}); use torch.distributed.run")
    share_gpu = a.backend == "gloo"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if share_gpu else dev        # where collectives run
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm

    B, K, Wm = a.batch, a.steps, a.warmup
    strong = a.frames > 0
    n_total = a.frames if strong else B * world                 # frames per step over all ranks
    f0, f1 = parallel.shard_range(n_total, rank, world)         # this rank's contiguous block of every step
    n_local = f1 - f0
    sds = synth.to_torch(synth.make_state_dicts(0))                     # random-init weights of the real architecture
    sw = can_swapper(type("Cfg", (), {"device_id": local_rank, "flag_force_cpu": False})(), state_dicts=sds, max_batch=B,
                     fp8_weights=a.fp8_weights, latency_mode=a.latency_mode)
    eng = sw.engine

    # one-time broadcast of the source identities (2 KB each); every rank derives T's modulated weights locally
    nid = max(1, min(a.identities, 8))
    sid = torch.from_numpy(synth.make_identity(7, n=nid)).to(cdev) if rank == 0 else torch.zeros(nid, 512, device=cdev)
    parallel.broadcast_identity(sid, src=0)
    sid = sid.to(dev)
    for k in range(nid):
        eng.set_identity(sid[k:k + 1], slot=k)
    frame_ids = sid[torch.arange(B, device=dev) % nid] if nid > 1 else None      # per-frame identity rows of one launch

    # synthetic inputs resident in HBM: a pool of 4 x B distinct frames.  Frame g of a step (global index) reads pool frame
    # (g + 131 * step) mod 4B, so the result of a frame does not depend on how many ranks share the job.
    P = 4 * B
    inp = synth.make_frame_inputs(P, seed=1000, size=256)
    pool = {k: torch.from_numpy(inp[k]).to(dev) for k in ("img", "x_t", "x_can")}
    out_u8 = torch.empty(max(n_local, 1), 512, 512, 3, dtype=torch.uint8, device=dev)

    def step(i, gather=None):
        """One step: this rank's block of the job in chunks of B frames; finished chunks go to rank 0 asynchronously."""
        for t0 in range(0, n_local, B):
            n = min(B, n_local - t0)
            idx = (torch.arange(f0 + t0, f0 + t0 + n, device=dev) + 131 * i) % P
            eng.swap_frames(pool["img"][idx], pool["x_t"][idx], pool["x_can"][idx], None if frame_ids is None else frame_ids[:n],
                            want_f32=False, want_u8=True, out_u8=out_u8[t0:t0 + n])
            if gather is not None:
                gather.push(out_u8[t0:t0 + n].to(cdev))
        return gather.finish() if gather is not None else None

    for i in range(Wm):
        step(i)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    sync()
    t0 = time.perf_counter()
    gathered = None
    for i in range(K):
        g = parallel.ChunkedFrameGather(n_total, B, device=cdev) if world > 1 else None     # chunked gather over xGMI, overlapped
        r = step(i, g)
        gathered = r if world > 1 else out_u8[:n_local]
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        assert gathered.shape[0] == n_total
        if a.dump_crc:
            import zlib
            fr = gathered.cpu().numpy()
            with open(a.dump_crc, "w") as f:
                json.dump([zlib.crc32(fr[k].tobytes()) for k in range(fr.shape[0])], f)

    # ---- roofline of the dominant kernel family (conv_igemm): HIP events around every launch, same workload
    prof = None
    if rank == 0:
        eng.profile_begin()
        for i in range(K):
            step(i)
        prof = eng.profile_end()
    if world > 1:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import canonswap_ref as O          # cpu_baseline leg: the oracle timed on this node's host cores
        cores = min(os.cpu_count() or 1, 32)           # threads used ("cores"); more than 32 slows PyTorch-CPU down on this path
        torch.set_num_threads(cores)
        inp = synth.make_frame_inputs(1, seed=1000, size=256)
        cargs = [torch.from_numpy(inp[k]) for k in ("img", "x_t", "x_can")]
        cid = torch.from_numpy(synth.make_identity(7))
        O.swap_frame(sds, *cargs, cid)                 # warm-up frame
        n_cpu, t1 = 8, time.perf_counter()          # about 11 s of CPU work
        for _ in range(n_cpu):
            O.swap_frame(sds, *cargs, cid)
        cdt = time.perf_counter() - t1
        cpu = {"value": round(n_cpu / cdt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
               "sample": f"{n_cpu} frames (1 warm-up), batch 1, fp32 PyTorch-CPU restatement of the same path (oracle/)"}

    traffic, traffic_src = None, None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    if rank == 0 and os.path.exists(tpath):      # PMC counters cannot be read from inside this process: committed rocprofv3 passes
        import json as _json
        t = _json.load(open(tpath))
        if int(t.get("batch", -1)) == B:
            traffic = t["fetch_bytes_per_conv_launch_x2"] + t["write_bytes_per_conv_launch_raw"]
            traffic_src = "profiles/hbm_traffic.json: " + t["source"]

    if rank == 0:
        frames = K * n_total
        fps = frames / dt
        conv_s = prof["conv_ms"] / 1e3
        achieved = prof["conv_flops"] / conv_s / 1e12
        line = {
            "metric": "frames/sec at 512x512 (generator hot path F->W->T->R->W->G)",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f16 activations x e4m3 weights (per-out-channel scale, expanded to f16 for the MFMA)" if a.fp8_weights else "f16",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[3]: 512x512 video of {n_total} frames sharded over {world} GPU(s) in contiguous "
                                    "blocks" if strong else "BASELINE configs[2]: 512x512 video, frames batched on each GPU") +
                                   " (256x256 crops in, random-init weights of the real architecture)",
                       "frames_per_step": n_total, "frames_per_launch_per_gpu": B, "frames_total": frames,
                       "parallelism": f"frame-shard x{world}", "identities_resident": nid, "latency_mode": bool(a.latency_mode),
                       "accumulate": "fp32", "debug_decodes": False},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_TFLOPS_F16, 4), "traffic": traffic, "traffic_unit": "HBM bytes per conv launch",
                         "traffic_source": traffic_src,
                         "kernel": "conv_halo + conv_igemm (every convolution launch)", "launches_per_step": prof["conv_launches"] // K,
                         "avg_launch_us": round(prof["conv_ms"] * 1e3 / prof["conv_launches"], 2),
                         "algorithmic_gflop_per_frame": round(prof["conv_flops"] / (K * n_local) / 1e9, 1),
                         # MFMA work actually issued (padded channel counts, phase-decomposed up-sampling convs): the utilisation of the
                         # matrix pipe, as opposed to the algorithmic rate above
                         "executed_tflops": round(prof["exec_flops"] / conv_s / 1e12, 2),
                         "executed_frac": round(prof["exec_flops"] / conv_s / 1e12 / PEAK_TFLOPS_F16, 4),
                         "other_kernels_ms_per_step": round((prof["other_ms"] + prof["warp_ms"]) / K, 3),
                         "conv_ms_per_step": round(prof["conv_ms"] / K, 3),
                         "end_to_end_frac": round(ALGO_GFLOP_PER_FRAME * 1e9 * fps / world / (PEAK_TFLOPS_F16 * 1e12), 4)},
            # the HBM-bound row of the path: trilinear feature warp (F.grid_sample, warping_network.py:46-47), fp32 volumes:
            # algorithmic bytes per frame and call = 8.39 MB in + 0.79 MB grid + 8.39 MB out (+ 4.19 MB fp16 copy on the first call)
            "warp_roofline": {"bound": "hbm", "kernel": "grid_sample_kernel",
                              "achieved": round((2 * 17.56e6 + 4.19e6) * n_local * K / (prof["warp_ms"] / 1e3) / 1e9, 1),
                              "peak": 8000.0, "unit": "GB/s",
                              "frac": round((2 * 17.56e6 + 4.19e6) * n_local * K / (prof["warp_ms"] / 1e3) / 8e12, 4),
                              "avg_launch_us": round(prof["warp_ms"] * 1e3 / max(prof["warp_launches"], 1), 2), "traffic": None},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
