"""conv_wide alone at the engine's T shapes (B x 64 x 64 x 512 -> 2 x 512): microseconds per launch by HIP events next to conv_halo's
128 x 256 tile and, with the instrumented build (python tools/build_variant.py wtl -DW_TL; CANONSWAP_LIB=ab/wtl.so), where a wave's cycles go.

    python tools/wide_probe.py [--batch 64] [--reps 10]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as ops  # noqa: E402
from canonswap_amd import _lib, pack  # noqa: E402

DEV = "cuda:0"
PH = ["startup", "ring prime", "vm wait", "barrier", "stage", "main loop", "next-item stage", "epilogue"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cfgs", default="17,31")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    B, Cc = a.batch, 512
    r = np.random.Generator(np.random.PCG64(3))
    lib = _lib.load()
    tl = None
    if hasattr(lib, "cs_debug_set_wide_tl"):
        cap = 1024
        tl = torch.zeros(cap * 12, dtype=torch.int64, device=DEV)
        lib.cs_debug_set_wide_tl.argtypes = [C.c_void_p, C.c_long]
        lib.cs_debug_set_wide_tl(C.c_void_p(tl.data_ptr()), cap)
    x = torch.relu(torch.randn(B, 1, 64, 64, Cc, device=DEV)).half()
    sc = 1.0 / np.sqrt(9 * Cc)
    w = (sc * r.standard_normal((2 * Cc, Cc, 3, 3))).astype(np.float32)
    wp = torch.from_numpy(pack.pack_conv(w, 2 * Cc)).to(DEV)
    bias = torch.randn(Cc, device=DEV) * 0.1
    m4 = torch.rand(B, 64, 64, 4, device=DEV)
    res = torch.randn(B, 1, 64, 64, Cc, device=DEV)
    o16 = torch.empty(B, 1, 64, 64, Cc, dtype=torch.float16, device=DEV)
    o32 = torch.empty(B, 1, 64, 64, Cc, dtype=torch.float32, device=DEV)
    s2 = torch.rand(Cc, device=DEV) + 0.5
    gfl = 2 * 9 * Cc * 2 * Cc * B * 4096 / 1e9
    wps = torch.from_numpy(pack.pack_conv(w[:Cc], Cc)).to(DEV)
    res16 = res.half()
    so = torch.empty(B * 4 * 8 * 2 * Cc * 2, dtype=torch.float32, device=DEV)
    actv = torch.relu(torch.randn(B, 1, 64, 64, 1536, device=DEV)).half()
    wsp = torch.from_numpy(pack.pack_conv((r.standard_normal((2 * Cc, 128, 3, 3)) / np.sqrt(9 * 128)).astype(np.float32), 2 * Cc)).to(DEV)
    stats = torch.stack([torch.zeros(B, Cc), torch.ones(B, Cc)], dim=2).contiguous().to(DEV)
    for cfg in [int(c) for c in a.cfgs.split(",")]:
        cases = {
            "T conv1 (blend, relu, fp16 out)": (1.0, lambda: ops.conv(x, wp, 2 * Cc, Cc, (1, 3, 3), bias=bias, pixscale=m4, ps_stride=4, act0="relu", out0=o16, mode=1, cfg=cfg)),
            "T conv2 (blend, fp32 res, fp32 + fp16 out)": (1.0, lambda: ops.conv(x, wp, 2 * Cc, Cc, (1, 3, 3), bias=bias, pixscale=m4, ps_stride=4, res=res, out0=o32,
                                                                                 s2=s2, t2=bias, act1="relu", out1=o16, mode=1, cfg=cfg)),
            "R c1 (lrelu, fp16 out)": (0.5, lambda: ops.conv(x, wps, Cc, Cc, (1, 3, 3), bias=bias, act0="lrelu", slope0=0.01, out0=o16, cfg=cfg)),
            "R c2 (fp32 res, fp32 + fp16 out)": (0.5, lambda: ops.conv(x, wps, Cc, Cc, (1, 3, 3), bias=bias, res=res, out0=o32, s2=s2, t2=bias, act1="lrelu", slope1=0.01,
                                                                       out1=o16, cfg=cfg)),
            "G c0 (fp16 out + statistics)": (0.5, lambda: ops.conv(x, wps, Cc, Cc, (1, 3, 3), bias=bias, out0=o16, stat_out=so, cfg=cfg)),
            "G c1 (fp16 res, fp16 out + statistics)": (0.5, lambda: ops.conv(x, wps, Cc, Cc, (1, 3, 3), bias=bias, res=res16, out0=o16, stat_out=so, cfg=cfg)),
        }
        # SPADE gamma / beta conv of G_middle (util.py:295-302): 128 -> 2 x 512 on a 128-channel slice of the 1536-channel actv buffer
        cases["G SPADE n0 (128 -> 2 x 512, IN-modulate x)"] = (0.25, lambda: ops.conv(actv[..., :128], wsp, 2 * Cc, Cc, (1, 3, 3), cin=128, bias=bias, bias2=bias,
                                                                                       res=res16, stats=stats, act0="lrelu", slope0=0.2, out0=o16, mode=2, cfg=cfg))
        for name, (fscale, fn) in cases.items():
            if a.only and a.only not in name:
                continue
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.reps
            print(f"cfg {cfg} {name:42s} {us:8.1f} us/launch  {fscale * gfl / us * 1e3:7.1f} TFLOP/s ({fscale * gfl / us * 1e3 / 2500:.3f} of peak)")
            if tl is not None and cfg == 31:
                tl.zero_()
                fn()
                torch.cuda.synchronize()
                t = tl.view(-1, 12).cpu().numpy()
                t = t[t[:, 8] > 0]
                life = t[:, :8].sum(axis=1)
                print(f"   waves {len(t)}, items/wave {t[:, 8].mean():.1f}, chunks {t[:, 9].mean():.0f}, life {life.mean():.0f} cycles (min {life.min()}, max {life.max()})")
                print("   " + "  ".join(f"{PH[i]} {t[:, i].mean():.0f} ({t[:, i].mean() / life.mean():.1%})" for i in range(8)))
                print("   per chunk: " + "  ".join(f"{PH[i]} {t[:, i].mean() / t[:, 9].mean():.0f}" for i in (2, 3, 4, 5)) +
                      "   per item: " + "  ".join(f"{PH[i]} {t[:, i].mean() / t[:, 8].mean():.0f}" for i in (1, 6, 7)))


if __name__ == "__main__":
    main()
