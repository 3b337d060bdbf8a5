"""Experiment (GPU box): the whole per-frame body replayed as a HIP graph (captured through torch.cuda.CUDAGraph) vs eager launches."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from canonswap_amd import synth  # noqa: E402
from canonswap_amd.can_swap_e2e import can_swapper  # noqa: E402

dev = torch.device("cuda", 0)
sds = synth.to_torch(synth.make_state_dicts(0))
sid = torch.from_numpy(synth.make_identity(7)).to(dev)
for B in [int(b) for b in (sys.argv[1:] or ["1", "4", "16", "32"])]:
    sw = can_swapper(type("Cfg", (), {"device_id": 0, "flag_force_cpu": False})(), state_dicts=sds, max_batch=B)
    eng = sw.engine
    eng.set_identity(sid)
    inp = synth.make_frame_inputs(B, seed=1000, size=256)
    a = [torch.from_numpy(inp[k]).to(dev) for k in ("img", "x_t", "x_can")]
    out = torch.empty(B, 512, 512, 3, dtype=torch.uint8, device=dev)

    def step():
        eng.swap_frames(*a, want_f32=False, want_u8=True, out_u8=out)

    def timeit(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    n = 50 if B <= 4 else 15
    te = timeit(step, n)
    ref = out.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.synchronize()
    out.zero_()
    tg = timeit(g.replay, n)
    same = bool((out == ref).all())
    print(f"B={B}: eager {te * 1e3:.2f} ms ({B / te:.1f} fps)   graph {tg * 1e3:.2f} ms ({B / tg:.1f} fps)   identical={same}", flush=True)
    del sw, eng
