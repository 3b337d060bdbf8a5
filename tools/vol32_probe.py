"""vol32 kernel alone at the engine's shapes (B x 64 x 64 x 16 x 32): microseconds per launch by HIP events and, with the instrumented
build (python tools/build_variant.py v32tl -DV32_TL; CANONSWAP_LIB=ab/v32tl.so), where a wave's cycles go.

    python tools/vol32_probe.py [--batch 32] [--reps 20]
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
from canonswap_amd import _lib  # noqa: E402

DEV = "cuda:0"
PH = ["startup", "prologue wait", "DMA issue", "stores", "main loop", "epilogue", "vm wait", "barrier", "tail"]


def vol(N, C=32, dtype=torch.float16, rnd=None, relu=False):
    t = torch.empty(N, 64, 64, 16, C, dtype=dtype, device=DEV)
    if rnd is not None:
        x = torch.from_numpy(rnd.standard_normal(t.shape).astype(np.float32))
        t.copy_((torch.relu(x) if relu else x).to(DEV))
    return t.permute(0, 3, 1, 2, 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cfg", type=int, default=30)
    a = ap.parse_args()
    B = a.batch
    r = np.random.Generator(np.random.PCG64(3))
    lib = _lib.load()
    tl = None
    if hasattr(lib, "cs_debug_set_vol32_tl"):
        cap = 4096
        tl = torch.zeros(cap * 12, dtype=torch.int64, device=DEV)
        lib.cs_debug_set_vol32_tl.argtypes = [C.c_void_p, C.c_long]
        lib.cs_debug_set_vol32_tl(C.c_void_p(tl.data_ptr()), cap)
    w = torch.from_numpy((r.standard_normal((27, 32, 32)) * 0.02).astype(np.float16)).to(DEV)
    w3 = torch.from_numpy((r.standard_normal((81, 32, 32)) * 0.02).astype(np.float16)).to(DEV)
    b = torch.from_numpy(r.standard_normal(32).astype(np.float32)).to(DEV)
    x = vol(B, rnd=r, relu=True)
    x2 = vol(B, C=64, rnd=r)
    res = vol(B, dtype=torch.float32, rnd=r)
    kw = dict(tile=(4, 4)) if a.cfg == 12 else {}
    cases = {
        "c1 (fp16 out, relu)": lambda: ops.conv(x, w, 32, 32, (3, 3, 3), bias=b, act0="relu", out0=vol(B), cfg=a.cfg, **kw),
        "c2 (res, fp32 + fp16 out)": lambda: ops.conv(x, w, 32, 32, (3, 3, 3), bias=b, res=res, out0=vol(B, dtype=torch.float32), s2=b, t2=b,
                                                       act1="relu", out1=vol(B), cfg=a.cfg, **kw),
        "split + stats": lambda: ops.conv(x2, w3, 32, 32, (3, 3, 3), cin=96, bias=b, out0=vol(B, dtype=torch.float32), cfg=a.cfg, hilo=True,
                                          stat_out=torch.empty(B * 256 * 64, dtype=torch.float32, device=DEV) if a.cfg == 30 else None, **kw),
    }
    # the fused ResBlock3d kernel (vol32_fused.hip)
    import ctypes as CC
    tlf = None
    if hasattr(lib, "cs_debug_set_vol32f_tl"):
        tlf = torch.zeros(4096 * 12, dtype=torch.int64, device=DEV)
        lib.cs_debug_set_vol32f_tl.argtypes = [CC.c_void_p, CC.c_long]
        lib.cs_debug_set_vol32f_tl(CC.c_void_p(tlf.data_ptr()), 4096)
    xa = torch.relu(torch.randn(B, 64, 64, 16, 32, device=DEV)).half()
    xr = torch.randn(B, 64, 64, 16, 32, device=DEV)
    o0 = torch.empty_like(xr); o1 = torch.empty_like(xa)
    w2 = torch.from_numpy((r.standard_normal((27, 32, 32)) * 0.02).astype(np.float16)).to(DEV)
    pp = lambda t: CC.c_void_p(t.data_ptr())
    stream = lambda: CC.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fused():
        _lib.check(lib.cs_op_resblock3d(pp(xa), pp(xr), pp(o0), pp(o1), B, 64, 64, pp(w), pp(w2), pp(b), pp(b), pp(b), pp(b), 1, 0.0, stream()), "resblock")

    if a.cfg == 30:
        cases["fused ResBlock3d (c1 + c2)"] = fused
    FPH = ["startup", "prologue", "compute", "memory instr", "epilogue", "lgkm wait", "vm wait", "barrier", "step head", "tail"]
    # transform staging (GroupNorm apply + residual + write-back inside the staging of the split-precision conv)
    yv = vol(B, dtype=torch.float32, rnd=r); rv = vol(B, dtype=torch.float32, rnd=r)
    stats = torch.stack([torch.zeros(B, 32), torch.ones(B, 32)], dim=2).contiguous().to(DEV)
    gam = torch.ones(32, device=DEV); bet = torch.zeros(32, device=DEV)
    if a.cfg == 30:
        sto = torch.empty(B * 256 * 64, dtype=torch.float32, device=DEV)
        xo = vol(B, dtype=torch.float32)
        cases["split + stats, XF kind 1"] = lambda: ops.conv(x2, w3, 32, 32, (3, 3, 3), cin=96, bias=b, out0=vol(B, dtype=torch.float32), cfg=30, hilo=True,
                                                              stat_out=sto, xf=dict(kind=1, y=yv))
        cases["split + stats, XF kind 2"] = lambda: ops.conv(x2, w3, 32, 32, (3, 3, 3), cin=96, bias=b, out0=vol(B, dtype=torch.float32), cfg=30, hilo=True,
                                                              stat_out=sto, xf=dict(kind=2, y=yv, stats=stats, gamma=gam, beta=bet, slope=0.01))
        cases["split + stats, XF kind 2 + res + write-back"] = lambda: ops.conv(x2, w3, 32, 32, (3, 3, 3), cin=96, bias=b, out0=vol(B, dtype=torch.float32),
                                                                                cfg=30, hilo=True, stat_out=sto,
                                                                                xf=dict(kind=2, y=yv, stats=stats, gamma=gam, beta=bet, slope=0.01, res=rv, out=xo))
    gfl = 2 * 27 * 32 * 32 * B * 65536 / 1e9
    for name, fn in cases.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        outs = [fn for _ in range(a.reps)]
        e0.record()
        for f in outs:
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        print(f"{name:44s} {us:8.1f} us/launch  {gfl / us * 1e-3:7.1f} TFLOP/s algorithmic ({gfl / us * 1e-3 / 2500:.3f} of peak)")
        if tlf is not None and name.startswith("fused"):
            tlf.zero_()
            fn()
            torch.cuda.synchronize()
            t = tlf.view(-1, 12).cpu().numpy()
            t = t[t[:, 10] > 0]
            for role in (0, 1):
                q = t[t[:, 11] == role]
                life = q[:, :10].sum(axis=1)
                print(f"   role {'conv2' if role else 'conv1'}: waves {len(q)}, steps {q[:, 10].mean():.1f}, life {life.mean():.0f} cycles")
                print("      per step: " + "  ".join(f"{FPH[i]} {q[:, i].mean() / q[:, 10].mean():.0f}" for i in range(2, 9)))
            continue
        if tl is not None and not name.startswith("fused"):
            tl.zero_()
            fn()
            torch.cuda.synchronize()
            t = tl.view(-1, 12).cpu().numpy()
            t = t[t[:, 9] > 0]
            life = t[:, :9].sum(axis=1)
            print(f"   waves {len(t)}, steps/wave {t[:, 9].mean():.1f}, life {life.mean():.0f} cycles (min {life.min()}, max {life.max()})")
            print("   " + "  ".join(f"{PH[i]} {t[:, i].mean():.0f} ({t[:, i].mean() / life.mean():.1%})" for i in range(9)))
            print("   per step: " + "  ".join(f"{PH[i]} {t[:, i].mean() / t[:, 9].mean():.0f}" for i in range(2, 8)))


if __name__ == "__main__":
    main()
