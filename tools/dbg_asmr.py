import sys, os, zlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import hip_ops as ops
DEV = "cuda:0"
out_path = sys.argv[1]
res = {}
for k, tile in (((3, 3, 3), (8, 8)), ((7, 7, 1), (2, 8))):
    r = np.random.Generator(np.random.PCG64(5))
    N, D, H, W, cin, cin_real, cout_pad = 3, 16, 16, 16, 144, 142, 160
    x = torch.from_numpy(r.standard_normal((N, D, H, W, cin_real)).astype(np.float16))
    w = (r.standard_normal((150, cin_real, *k)) / np.sqrt(cin_real * np.prod(k))).astype(np.float32)
    buf = torch.zeros(N, D, H, W, cin, dtype=torch.float16, device=DEV); buf[..., :cin_real] = x.to(DEV)
    for ragged in (False, True):
        wp = ops.packed_weight(torch.from_numpy(w), cout_pad, DEV)
        if ragged: ops.pair_ragged(wp, cout_pad, cin, k)
        out = torch.zeros(N, D, H, W, cout_pad, dtype=torch.float32, device=DEV)
        ops.conv(buf, wp, cout_pad, cout_pad, k, cin=cin, out0=out, cfg=19, tile=tile, ragged=ragged)
        torch.cuda.synchronize()
        res[f"{k}_{ragged}"] = out.cpu().numpy()
np.savez(out_path, **res)
