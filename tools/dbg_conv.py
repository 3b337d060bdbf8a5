"""Debug: conv_halo vs conv_igemm on the GPU for a matrix of shapes (prints relative differences)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as ops
DEV = "cuda:0"

def run(N, Cin, Cout, D, H, W, k, hcfg=-2, tile=(0, 0), ck=0):
    r = np.random.Generator(np.random.PCG64(1))
    x = torch.from_numpy(r.standard_normal((N, D, H, W, Cin)).astype(np.float16)).to(DEV)
    nch = (Cin + 31) // 32
    wp = torch.from_numpy((r.standard_normal((nch * int(np.prod(k)), Cout, 32)) * 0.05).astype(np.float16)).to(DEV)
    outs = []
    for cfg, tl in ((-1, (0, 0)), (hcfg, tile)):
        out = torch.zeros(N, D, H, W, Cout, dtype=torch.float32, device=DEV)
        ops.conv(x, wp, Cout, Cout, k, cin=Cin, out0=out, cfg=cfg, tile=tl, ck=ck)
        torch.cuda.synchronize()
        outs.append(out)
    a, b = outs
    d = (a - b).abs()
    bad_n = [float(d[n].max()) for n in range(N)]
    return float((a - b).norm() / (a.norm() + 1e-9)), bad_n

cases = []
for N in (1, 2, 3):
    for Cin in (64, 96, 128, 160):
        for Cout in (128, 256):
            cases.append((N, Cin, Cout, 1, 16, 16, (1, 3, 3)))
cases += [(2, 144, 192, 4, 16, 16, (3, 3, 3)), (2, 144, 192, 16, 64, 64, (3, 3, 3)), (1, 144, 64, 16, 64, 64, (3, 3, 3)),
          (2, 112, 64, 16, 32, 32, (3, 3, 3))]
for c in cases:
    for ck in ((32, 64) if c[6][0] == 1 and c[1] % 64 == 0 else (32,)):
        try:
            e, per_n = run(*c, ck=ck)
            print(c, "ck", ck, "rel %.2e" % e, ["%.1e" % v for v in per_n], flush=True)
        except Exception as ex:
            print(c, "ck", ck, "EXC", str(ex)[:100], flush=True)
