"""conv_lat at one frame (1 x 64 x 64 x 512): microseconds per launch next to conv_halo and, with the instrumented build
(python tools/build_variant.py lattl -DLAT_TL; CANONSWAP_LIB=ab/lattl.so), where a wave's cycles go.   python tools/lat_probe.py"""
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
PH = ["startup", "head wait", "barrier", "stage", "K-steps", "reduction", "epilogue"]


def main():
    B, Cc = 1, 512
    r = np.random.Generator(np.random.PCG64(3))
    lib = _lib.load()
    tl = None
    if hasattr(lib, "cs_debug_set_lat_tl"):
        cap = 256 * 12
        tl = torch.zeros(cap * 8, dtype=torch.int64, device=DEV)
        lib.cs_debug_set_lat_tl.argtypes = [C.c_void_p, C.c_long]
        lib.cs_debug_set_lat_tl(C.c_void_p(tl.data_ptr()), cap)
    x = torch.relu(torch.randn(B, 1, 64, 64, Cc, device=DEV)).half()
    sc = 1.0 / np.sqrt(9 * Cc)
    w = (sc * r.standard_normal((2 * Cc, Cc, 3, 3))).astype(np.float32)
    wp = torch.from_numpy(pack.pack_conv(w, 2 * Cc)).to(DEV)
    wps = torch.from_numpy(pack.pack_conv(w[:Cc], Cc)).to(DEV)
    bias = torch.randn(Cc, device=DEV) * 0.1
    m4 = torch.rand(B, 64, 64, 4, device=DEV)
    res = torch.randn(B, 1, 64, 64, Cc, device=DEV)
    o16 = torch.empty(B, 1, 64, 64, Cc, dtype=torch.float16, device=DEV)
    o32 = torch.empty(B, 1, 64, 64, Cc, dtype=torch.float32, device=DEV)
    s2 = torch.rand(Cc, device=DEV) + 0.5
    so = torch.empty(B * 4 * 8 * 2 * Cc * 2, dtype=torch.float32, device=DEV)
    gfl = 2 * 9 * Cc * 2 * Cc * B * 4096 / 1e9
    for cfg in (32, 10, 11):
        cases = {
            "T conv1 (blend, relu, fp16 out)": (1.0, lambda: ops.conv(x, wp, 2 * Cc, Cc, (1, 3, 3), bias=bias, pixscale=m4, ps_stride=4, act0="relu", out0=o16, mode=1, cfg=cfg)),
            "T conv2 (blend, fp32 res, fp32 + fp16 out)": (1.0, lambda: ops.conv(x, wp, 2 * Cc, Cc, (1, 3, 3), bias=bias, pixscale=m4, ps_stride=4, res=res, out0=o32,
                                                                                 s2=s2, t2=bias, act1="relu", out1=o16, mode=1, cfg=cfg)),
            "R c1 (lrelu, fp16 out)": (0.5, lambda: ops.conv(x, wps, Cc, Cc, (1, 3, 3), bias=bias, act0="lrelu", slope0=0.01, out0=o16, cfg=cfg)),
            "G c0 (fp16 out + statistics)": (0.5, lambda: ops.conv(x, wps, Cc, Cc, (1, 3, 3), bias=bias, out0=o16, stat_out=so, cfg=cfg)),
        }
        for name, (fscale, fn) in cases.items():
            if (cfg == 10 and not name.startswith("T")) or (cfg == 11 and name.startswith("T")):
                continue
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            print(f"cfg {cfg} {name:42s} {us:8.1f} us/launch  {fscale * gfl / us * 1e3:7.1f} TFLOP/s ({fscale * gfl / us * 1e3 / 2500:.3f} of peak)")
            if tl is not None and cfg == 32:
                tl.zero_()
                fn()
                torch.cuda.synchronize()
                t = tl.view(-1, 8).cpu().numpy().reshape(256, 12, 8)
                ng = 3 if name.startswith("T") else 6
                for g in range(ng):
                    u = t[:, g * (12 // ng):(g + 1) * (12 // ng), :].reshape(-1, 8).astype(np.float64)
                    life = u[:, :7].sum(axis=1)
                    print(f"   K-group {g}: life {life.mean():.0f} cycles; " + "  ".join(f"{PH[i]} {u[:, i].mean():.0f} ({u[:, i].mean() / life.mean():.0%})" for i in range(7)) +
                          "   per chunk: " + "  ".join(f"{PH[i]} {u[:, i].mean() / 8:.0f}" for i in (1, 2, 3, 4)))

if __name__ == "__main__":
    main()
