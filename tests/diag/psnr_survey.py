"""GPU box: PSNR of the engine's frames against the fp32 CPU oracle over many frames of bench.py's input pool (seed 1000) - the distribution behind
the two-frame `psnr_db_min` of the bench line.   python tests/diag/psnr_survey.py [first] [count] [stride]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from canonswap_amd import synth  # noqa: E402
from canonswap_amd.can_swap_e2e import can_swapper  # noqa: E402
from oracle import canonswap_ref as O  # noqa: E402

first, count, stride = (int(x) for x in (sys.argv[1:4] + ["0", "16", "4"][len(sys.argv) - 1:]))
torch.set_num_threads(min(os.cpu_count() or 1, 32))
sds = synth.to_torch(synth.make_state_dicts(0))
sw = can_swapper(None, state_dicts=sds, max_batch=8)
inp = synth.make_frame_inputs(256, seed=1000, size=256)
idv = torch.from_numpy(synth.make_identity(7))
res = []
for j in range(first, first + count * stride, stride):
    a = [torch.from_numpy(inp[k][j:j + 1]) for k in ("img", "x_t", "x_can")]
    with torch.no_grad():
        ref = O.swap_frame(sds, *a, idv, debug=True)
    r = sw.swap_frames(a[0].cuda(), a[1].cuda(), a[2].cuda(), idv.cuda(), debug=True)
    p = O.psnr(r["out"].cpu(), ref["out"])
    # where the error enters: each stage fed the oracle's input
    f_s = sw.extract_feature_3d(a[0].cuda())
    f_can, _ = sw.warping_module.warp(ref["f_s"].cuda(), a[1].cuda(), a[2].cuda())
    f_swap = sw.swap_module(ref["f_can"].cuda(), idv.cuda())
    f_ref = sw.refine_module(ref["f_swap"].cuda())
    wd = sw.warp_decode(ref["f_ref"].cuda(), a[2].cuda(), a[1].cuda())
    rel = lambda x, y: float((x.detach().double().cpu() - y.double()).norm() / y.double().norm())
    res.append(p)
    print(f"frame {j:3d}: {p:6.2f} dB   rec_can {O.psnr(r['rec_can'].cpu(), ref['rec_can']):6.2f}  swap_can {O.psnr(r['swap_can'].cpu(), ref['swap_can']):6.2f}  "
          f"| stage errors F {rel(f_s, ref['f_s']):.1e} W {rel(f_can, ref['f_can']):.1e} T {rel(f_swap, ref['f_swap']):.1e} R {rel(f_ref, ref['f_ref']):.1e} "
          f"W.forward+G from oracle f_ref {O.psnr(wd['out'].cpu(), ref['out']):6.2f} dB  | out mean {float(ref['out'].mean()):.3f} std {float(ref['out'].std()):.3f}", flush=True)
print(f"min {min(res):.2f}  median {float(np.median(res)):.2f}  max {max(res):.2f} dB over {len(res)} frames")
