"""One-frame 512-channel 3x3 convs: the same launch with WARM weights (one weight set, re-used: it sits in L2 / Infinity Cache) and with COLD
weights (a rotation over more sets than the 256 MB Infinity Cache holds: every set comes from HBM, as in the one-frame step, where every
conv's weights are read once per frame).  GPU box:  python tools/lat_warm_cold.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as ops  # noqa: E402

DEV = "cuda:0"


def bench(name, cout_pad, cout, mode, cfg, nsets, prefetch=False):
    r = np.random.Generator(np.random.PCG64(1))
    N, H, W, Cin = 1, 64, 64, 512
    x = torch.from_numpy(np.maximum(r.standard_normal((N, 1, H, W, Cin)), 0).astype(np.float16)).to(DEV)
    w0 = torch.from_numpy((r.standard_normal((16 * 9, cout_pad, 32)) * 0.02).astype(np.float16)).to(DEV)
    sets = [w0.clone() for _ in range(nsets)]
    out = torch.empty(N, 1, H, W, cout, dtype=torch.float16, device=DEV)
    m4 = torch.rand(N, H, W, 4, device=DEV)
    bias = torch.zeros(cout, device=DEV)
    kw = dict(bias=bias, act0="relu", out0=out, cfg=cfg, mode=mode)
    if mode == 1:
        kw.update(pixscale=m4, ps_stride=4)
    for i in range(min(nsets, 4)):
        ops.conv(x, sets[i], cout_pad, cout, (1, 3, 3), **kw)
    torch.cuda.synchronize()
    reps = max(40, nsets)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    sink = torch.zeros(1, device=DEV)
    ev[0].record()
    for i in range(reps):
        if prefetch:      # touch the NEXT set (a plain read of it) before this conv: what a prefetch kernel would do
            sink += sets[(i + 1) % nsets].view(-1)[::32].float().sum()
        ops.conv(x, sets[i % nsets], cout_pad, cout, (1, 3, 3), **kw)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"{name:44s} sets {nsets:3d} ({nsets * w0.numel() * 2 / 1e6:7.1f} MB){' +touch' if prefetch else '       '}  {ev[0].elapsed_time(ev[1]) / reps * 1e3:7.1f} us per launch")


if __name__ == "__main__":
    for cfg, cn in ((32, "conv_lat"), (10, "conv_halo 128x128")):
        bench(f"T blend 512 -> 2 x 512, {cn}", 1024, 512, 1, cfg, 1)
        bench(f"T blend 512 -> 2 x 512, {cn}", 1024, 512, 1, cfg, 40)
    for cfg, cn in ((32, "conv_lat"), (11, "conv_halo 128x64")):
        bench(f"512 -> 512, {cn}", 512, 512, 0, cfg, 1)
        bench(f"512 -> 512, {cn}", 512, 512, 0, cfg, 80)
