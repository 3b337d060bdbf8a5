"""GPU box: PSNR against the fp32 CPU oracle of EVERY frame of bench.py's input pool (4 x n frames of seed 1000, identity make_identity(7)[0];
n-frame launches as the bench runs them; a frame's bits do not depend on the launch it rides in, tests/test_gpu_batch32.py)
-> profiles/psnr_worst_frame.json, which bench.py reads to put the worst frame of the whole pool into its parity sample (VERDICT r3 item 7,
r4 item 7).   python tests/diag/psnr_pool.py [n = 64] [out.json] [launches = 4]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from canonswap_amd import synth  # noqa: E402
from canonswap_amd.can_swap_e2e import can_swapper  # noqa: E402
from oracle import canonswap_ref as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "psnr_worst_frame.json")
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.set_num_threads(min(os.cpu_count() or 1, 32))
sds = synth.to_torch(synth.make_state_dicts(0))
sw = can_swapper(None, state_dicts=sds, max_batch=n)
inp = synth.make_frame_inputs(4 * n, seed=1000, size=256)          # the bench's pool: 4 x batch frames
idv = torch.from_numpy(synth.make_identity(7))
ps = []
for l in range(launches):
    a = [torch.from_numpy(inp[k][l * n:(l + 1) * n]) for k in ("img", "x_t", "x_can")]
    got = sw.swap_frames(a[0].cuda(), a[1].cuda(), a[2].cuda(), idv.cuda())["out"].cpu()
    for j in range(n):
        with torch.no_grad():
            ref = O.swap_frame(sds, *(t[j:j + 1] for t in a), idv)["out"]
        ps.append(float(O.psnr(got[j:j + 1], ref)))
        print(f"frame {l * n + j:3d}: {ps[-1]:6.2f} dB", flush=True)
w = int(np.argmin(ps))
w0 = int(np.argmin(ps[:n]))
print(f"min {min(ps):.2f} dB at pool frame {w}   median {float(np.median(ps)):.2f}   max {max(ps):.2f} dB over {len(ps)} frames "
      f"(first launch: min {ps[w0]:.2f} dB at frame {w0})")
json.dump({"batch": n, "pool_seed": 1000, "pool_frames": len(ps), "worst_frame": w, "worst_frame_first_launch": w0,
           "psnr_db": [round(p, 2) for p in ps], "min_db": round(min(ps), 2),
           "median_db": round(float(np.median(ps)), 2), "source": "tests/diag/psnr_pool.py"}, open(out, "w"))
