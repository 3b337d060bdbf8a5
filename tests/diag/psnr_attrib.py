"""GPU box: which stage's error costs the frame its PSNR.  The engine's stages are chained through the class API with the oracle's tensor substituted
at one boundary at a time: PSNR of the final image when everything up to that boundary is exact.   python tests/diag/psnr_attrib.py 63 3 19 23"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from canonswap_amd import synth  # noqa: E402
from canonswap_amd.can_swap_e2e import can_swapper  # noqa: E402
from oracle import canonswap_ref as O  # noqa: E402

torch.set_num_threads(min(os.cpu_count() or 1, 32))
sds = synth.to_torch(synth.make_state_dicts(0))
sw = can_swapper(None, state_dicts=sds, max_batch=4)
inp = synth.make_frame_inputs(256, seed=1000, size=256)
idv = torch.from_numpy(synth.make_identity(7))
for j in [int(x) for x in (sys.argv[1:] or ["63", "3", "23"])]:
    a = [torch.from_numpy(inp[k][j:j + 1]) for k in ("img", "x_t", "x_can")]
    with torch.no_grad():
        ref = O.swap_frame(sds, *a, idv, debug=True)
    img, x_t, x_can, idc = a[0].cuda(), a[1].cuda(), a[2].cuda(), idv.cuda()

    def rest(f_s=None, f_can=None, f_swap=None, f_ref=None):
        if f_ref is None:
            if f_swap is None:
                if f_can is None:
                    if f_s is None:
                        f_s = sw.extract_feature_3d(img)
                    f_can, _ = sw.warping_module.warp(f_s, x_t, x_can)
                f_swap = sw.swap_module(f_can, idc)
            f_ref = sw.refine_module(f_swap)
        return O.psnr(sw.warp_decode(f_ref, x_can, x_t)["out"].cpu(), ref["out"])

    print(f"frame {j:3d}: all engine {rest():6.2f} | exact f_s {rest(f_s=ref['f_s'].cuda()):6.2f} | exact f_can {rest(f_can=ref['f_can'].cuda()):6.2f} | "
          f"exact f_swap {rest(f_swap=ref['f_swap'].cuda()):6.2f} | exact f_ref {rest(f_ref=ref['f_ref'].cuda()):6.2f} dB", flush=True)
