"""GPU box: PSNR of the one-frame latency mode against the fp32 oracle on a sample of the bench's pool frames, the known worst frame (63)
among them, next to the batched path's value for the same frame.   python tests/diag/psnr_latency.py [frames ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from canonswap_amd import synth  # noqa: E402
from canonswap_amd.can_swap_e2e import can_swapper  # noqa: E402
from oracle import canonswap_ref as O  # noqa: E402

torch.set_num_threads(16)
frames = [int(a) for a in sys.argv[1:]] or [0, 3, 63, 100, 170, 255]
sds = synth.to_torch(synth.make_state_dicts(0))
inp = synth.make_frame_inputs(256, seed=1000, size=256)
idv = torch.from_numpy(synth.make_identity(7))
lat = can_swapper(None, state_dicts=sds, max_batch=1, latency_mode=True)
std = can_swapper(None, state_dicts=sds, max_batch=1)
for j in frames:
    a = [torch.from_numpy(inp[k][j:j + 1]) for k in ("img", "x_t", "x_can")]
    with torch.no_grad():
        ref = O.swap_frame(sds, *a, idv)["out"]
    g = [t.cuda() for t in a]
    pl = O.psnr(lat.swap_frames(*g, idv.cuda())["out"].cpu(), ref)
    ps = O.psnr(std.swap_frames(*g, idv.cuda())["out"].cpu(), ref)
    print(f"pool frame {j:3d}: latency mode {pl:6.2f} dB   default mode {ps:6.2f} dB", flush=True)
