"""GPU box: the measured error of every stage gate of tests/test_gpu_stages.py (relative L2 / PSNR against the fp32 CPU oracle), for three
input seeds - the numbers the gates in that file are set from.   python tests/diag/stage_errors.py > gpurun_out/stage_errors.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from canonswap_amd import synth  # noqa: E402
from canonswap_amd.can_swap_e2e import can_swapper  # noqa: E402
from oracle import canonswap_ref as O  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


torch.set_num_threads(min(os.cpu_count() or 1, 32))
sds = synth.to_torch(synth.make_state_dicts(0))
sw = can_swapper(None, state_dicts=sds, max_batch=4)
for seed in (1000, 2000, 3000):
    inp = synth.make_frame_inputs(2, seed=seed, size=256)
    idv = torch.from_numpy(synth.make_identity(7 + seed // 1000))
    a = {k: torch.from_numpy(v) for k, v in inp.items()}
    with torch.no_grad():
        ref = O.swap_frame(sds, a["img"], a["x_t"], a["x_can"], idv, debug=True)
    f = sw.extract_feature_3d(a["img"].cuda())
    f_can, occ = sw.warping_module.warp(ref["f_s"].cuda(), a["x_t"].cuda(), a["x_can"].cuda())
    f_swap = sw.swap_module(ref["f_can"].cuda(), idv.cuda())
    f_ref = sw.refine_module(ref["f_swap"].cuda())
    ret = sw.warp_decode(ref["f_ref"].cuda(), a["x_can"].cuda(), a["x_t"].cuda())
    seg = sw.warping_module.warp_out(ref["f_ref"].cuda(), ref["occ2"].cuda())
    with torch.no_grad():
        seg_ref = O.warp_out(sds["warping_module"], ref["f_ref"], ref["occ2"])
    r = sw.swap_frames(a["img"].cuda(), a["x_t"].cuda(), a["x_can"].cuda(), idv.cuda(), debug=True, want_u8=True)
    print(f"seed {seed}: f_s {rel(f, ref['f_s']):.2e}  occ {rel(occ, ref['occ']):.2e}  f_can {rel(f_can, ref['f_can']):.2e}  "
          f"f_swap {rel(f_swap, ref['f_swap']):.2e}  f_ref {rel(f_ref, ref['f_ref']):.2e}  deformation {rel(ret['deformation'], ref['deformation']):.2e}  "
          f"occ2 {rel(ret['occlusion_map'], ref['occ2']):.2e}  seg {rel(seg, seg_ref):.2e}  "
          f"warp_decode out {O.psnr(ret['out'].cpu(), ref['out']):.2f} dB  frame out {O.psnr(r['out'].cpu(), ref['out']):.2f} dB  "
          f"rec_can {O.psnr(r['rec_can'].cpu(), ref['rec_can']):.2f} dB  swap_can {O.psnr(r['swap_can'].cpu(), ref['swap_can']):.2f} dB", flush=True)
