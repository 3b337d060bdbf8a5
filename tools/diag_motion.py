"""GPU box: accuracy and timing of the motion extractor M on the HIP engine vs the oracle."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from canonswap_amd import synth  # noqa: E402
from canonswap_amd.can_swap_e2e import can_swapper  # noqa: E402
from oracle import canonswap_ref as O  # noqa: E402

B = int(os.environ.get("B", "32"))
sds = synth.to_torch(synth.make_state_dicts(0, modules=synth.MODULES + ("motion_extractor",)))
sw = can_swapper(None, state_dicts=sds, max_batch=B)
img = torch.from_numpy(synth.make_smooth_images(4, seed=2000))
with torch.no_grad():
    ref = O.motion_extractor(sds["motion_extractor"], img)
out = sw.motion_extractor(img.cuda())
for k in ref:
    print(f"{k:6s} max abs err {float((out[k].cpu() - ref[k]).abs().max()):.2e}   (|ref| max {float(ref[k].abs().max()):.2f})")
x = torch.from_numpy(synth.make_smooth_images(B, seed=3000)).cuda()
for _ in range(3):
    sw.motion_extractor(x)
torch.cuda.synchronize()
t = time.perf_counter()
n = 20
for _ in range(n):
    sw.motion_extractor(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
print(f"M: B={B}  {dt * 1e3:.2f} ms/batch  {B / dt:.0f} frames/s  ({11.6e9 * B / dt / 1e12:.1f} TFLOP/s algorithmic)")
sw.engine.profile_begin()
sw.motion_extractor(x)
p = sw.engine.profile_end()
print("profile:", {k: round(v, 3) if isinstance(v, float) else v for k, v in p.items()})
