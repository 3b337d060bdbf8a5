"""Helper of the knob tests (the engine reads its CANONSWAP_* knobs once per process, so a knob needs a process of its own):
python tests/run_frame.py OUT.pt LATENCY(0|1) FRAME [FRAME ...]  - pool frames (seed 1000, identity 7: the bench's pool) through
can_swapper.swap_frames one per call, float outputs saved to OUT.pt as {frame: tensor}."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonswap_amd import synth  # noqa: E402
from canonswap_amd.can_swap_e2e import can_swapper  # noqa: E402

out_path, lat, frames = sys.argv[1], bool(int(sys.argv[2])), [int(a) for a in sys.argv[3:]]
sds = synth.to_torch(synth.make_state_dicts(0))
inp = synth.make_frame_inputs(max(frames) + 1, seed=1000, size=256)
idv = torch.from_numpy(synth.make_identity(7)).cuda()
sw = can_swapper(None, state_dicts=sds, max_batch=1, latency_mode=lat)
res = {}
for j in frames:
    g = [torch.from_numpy(inp[k][j:j + 1]).cuda() for k in ("img", "x_t", "x_can")]
    res[j] = sw.swap_frames(*g, idv)["out"].cpu()
sw.engine.close()
torch.save(res, out_path)
