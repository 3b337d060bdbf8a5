"""GPU box: sha1 of the engine's outputs on seeded inputs (whole frames at B = 4 and the warp stage) - run under two builds
(CANONSWAP_LIB=...) and compare the lines: a change that claims "same bits" must print the same hashes."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from canonswap_amd import synth  # noqa: E402
from canonswap_amd.can_swap_e2e import can_swapper  # noqa: E402

sd = synth.to_torch(synth.make_state_dicts(0))
sw = can_swapper(None, state_dicts=sd, max_batch=4)
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_frame_inputs(4, seed=1000, size=256).items()}
idv = torch.from_numpy(synth.make_identity(7)).cuda()


def h(t):
    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


f = sw.extract_feature_3d(inp["img"])
print("F      ", h(f))
fc, occ = sw.warping_module.warp(f, inp["x_t"], inp["x_can"])
print("W.warp ", h(fc), h(occ))
fs = sw.swap_module(fc, idv)
print("T      ", h(fs))
fr = sw.refine_module(fs)
print("R      ", h(fr))
