"""Micro-benchmark of single conv layers through the operator-level C ABI (GPU box).
    python tools/bench_conv.py            # representative layer shapes, conv_igemm vs conv_halo"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as ops  # noqa: E402

DEV = "cuda:0"
# name, N, Cin, Cout_pad, D, H, W, k, up_shift, halo cfg, tile, real_flop_factor
SHAPES = [
    ("Tscan B2", 2, 512, 1024, 1, 64, 64, (1, 3, 3), 0, -2, (0, 0)),
    ("Tscan B4", 4, 512, 1024, 1, 64, 64, (1, 3, 3), 0, -2, (0, 0)),
    ("Tscan B8", 8, 512, 1024, 1, 64, 64, (1, 3, 3), 0, -2, (0, 0)),
    ("Tscan B12", 12, 512, 1024, 1, 64, 64, (1, 3, 3), 0, -2, (0, 0)),
    ("Tscan B16", 16, 512, 1024, 1, 64, 64, (1, 3, 3), 0, -2, (0, 0)),
    ("Tscan B24", 24, 512, 1024, 1, 64, 64, (1, 3, 3), 0, -2, (0, 0)),
    ("Tscan B32", 32, 512, 1024, 1, 64, 64, (1, 3, 3), 0, -2, (0, 0)),
    ("Tscan B64", 64, 512, 1024, 1, 64, 64, (1, 3, 3), 0, -2, (0, 0)),
    ("T fused 512->1024 3x3 @64 B16", 16, 512, 1024, 1, 64, 64, (1, 3, 3), 0, -2, (0, 0)),
    ("G conv 512->512 3x3 @64 B16", 16, 512, 512, 1, 64, 64, (1, 3, 3), 0, -2, (0, 0)),
    ("G gb 128->1024 3x3 @64 B16", 16, 128, 1024, 1, 64, 64, (1, 3, 3), 0, -2, (0, 0)),
    ("G gb 128->512 3x3 @128 B16", 16, 128, 512, 1, 128, 128, (1, 3, 3), 0, -2, (0, 0)),
    ("T fused 512->1024 3x3 @64 B16 (128x128)", 16, 512, 1024, 1, 64, 64, (1, 3, 3), 0, 10, (0, 0)),
    ("G conv 512->512 3x3 @64 B16 (128x128)", 16, 512, 512, 1, 64, 64, (1, 3, 3), 0, 10, (0, 0)),
    ("G conv 256->256 3x3 @128 B16", 16, 256, 256, 1, 128, 128, (1, 3, 3), 0, -2, (0, 0)),
    ("G conv 256->256 3x3 @128 B16 (128x128)", 16, 256, 256, 1, 128, 128, (1, 3, 3), 0, 10, (0, 0)),
    ("G gb 128->1024 3x3 @64 B16 (128x128)", 16, 128, 1024, 1, 64, 64, (1, 3, 3), 0, 10, (0, 0)),
    ("G gb 128->512 3x3 @128 B16 (128x128)", 16, 128, 512, 1, 128, 128, (1, 3, 3), 0, 10, (0, 0)),
    ("G gb 128->128 3x3 @256 B16 (128x128)", 16, 128, 128, 1, 256, 256, (1, 3, 3), 0, 10, (0, 0)),
    ("G gb 128->128 3x3 @256 B16 (128x64)", 16, 128, 128, 1, 256, 256, (1, 3, 3), 0, 11, (0, 0)),
    ("G shared 256->384 3x3 @256 B4", 4, 256, 384, 1, 256, 256, (1, 3, 3), 0, -2, (0, 0)),
    ("W maskp 144->160 7x7x1 B8", 8, 144, 160, 16, 64, 64, (7, 7, 1), 0, 18, (2, 8)),
    ("W tail 144->160 3x3x3 B8", 8, 144, 160, 16, 64, 64, (3, 3, 3), 0, 18, (8, 8)),
    ("W enc0 112->64 3x3x3 B8", 8, 112, 64, 16, 64, 64, (3, 3, 3), 0, -2, (8, 8)),
    ("W dec4 128->32 3x3x3 up B8", 8, 128, 32, 16, 64, 64, (3, 3, 3), 1, -2, (8, 8)),
    ("W dec1 1024->256 3x3x3 @8 up B16", 16, 1024, 256, 16, 8, 8, (3, 3, 3), 1, -2, (8, 8)),
    ("3D 32->32 3x3x3 hwdc-like B16", 16, 32, 32, 16, 64, 64, (3, 3, 3), 0, 12, (4, 4)),
]


def run(name, N, Cin, Cout, D, H, W, k, us, hcfg, tile, kern):
    r = np.random.Generator(np.random.PCG64(1))
    Hs, Ws = H >> us, W >> us
    x = torch.from_numpy(r.standard_normal((N, D, Hs, Ws, Cin)).astype(np.float16)).to(DEV)
    nch = (Cin + 31) // 32
    wp = torch.from_numpy((r.standard_normal((nch * int(np.prod(k)), Cout, 32)) * 0.02).astype(np.float16)).to(DEV)
    out = torch.empty(N, D, H, W, Cout, dtype=torch.float16, device=DEV)
    cfg = -1 if kern == "igemm" else hcfg
    args = dict(cin=Cin, act0="relu", out0=out, cfg=cfg, up_shift=us, out_dims=(N, D, H, W), tile=tile if kern == "halo" else (0, 0))
    for _ in range(2):
        ops.conv(x, wp, Cout, Cout, k, **args)
    torch.cuda.synchronize()
    n = 20
    t = time.perf_counter()
    for _ in range(n):
        ops.conv(x, wp, Cout, Cout, k, **args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    fl = 2.0 * N * D * H * W * Cout * Cin * np.prod(k)
    return dt * 1e3, fl / dt / 1e12


def main():
    only = os.environ.get("BENCH_ONLY")
    for s in SHAPES:
        if only and only not in s[0]:
            continue
        res = []
        for kern in (("igemm", "halo") if os.environ.get("BENCH_IGEMM") else ("halo",)):
            try:
                ms, tf = run(*s, kern)
                res.append("%s %7.3f ms %7.1f TF/s" % (kern, ms, tf))
            except Exception as e:  # noqa: BLE001
                res.append("%s ERR %s" % (kern, str(e)[:80]))
        print("%-36s | %s" % (s[0], " | ".join(res)), flush=True)


if __name__ == "__main__":
    main()
