"""GPU box: where the whole-frame error of the fp16 engine comes from, frame by frame.
For several (frame seed, identity seed) pairs: PSNR of the full path, of G alone (oracle seg in), of W.forward + G (oracle
f_refined in), and the relative L2 error of every stage fed with the oracle's inputs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from canonswap_amd import synth  # noqa: E402
from canonswap_amd.can_swap_e2e import can_swapper  # noqa: E402
from oracle import canonswap_ref as O  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    sds = synth.to_torch(synth.make_state_dicts(0))
    sw = can_swapper(None, state_dicts=sds, max_batch=1)
    cu = lambda t: t.cuda()
    pairs = [(1000, 7), (2024, 5), (2024, 7), (1000, 5), (4242, 7), (31337, 3)]
    if len(sys.argv) > 1:
        pairs = pairs[: int(sys.argv[1])]
    for fs, ids in pairs:
        inp = synth.make_frame_inputs(1, seed=fs, size=256)
        a = {k: torch.from_numpy(v) for k, v in inp.items()}
        idv = torch.from_numpy(synth.make_identity(ids))
        with torch.no_grad():
            ref = O.swap_frame(sds, a["img"], a["x_t"], a["x_can"], idv)
        full = O.psnr(sw.swap_frames(cu(a["img"]), cu(a["x_t"]), cu(a["x_can"]), cu(idv))["out"].cpu(), ref["out"])
        g_only = O.psnr(sw.spade_generator(feature=cu(ref["seg"])).cpu(), ref["out"])
        wd = sw.warp_decode(cu(ref["f_ref"]), cu(a["x_can"]), cu(a["x_t"]))
        wg = O.psnr(wd["out"].cpu(), ref["out"])
        f_s = sw.extract_feature_3d(cu(a["img"]))
        f_can, occ = sw.warping_module.warp(cu(ref["f_s"]), cu(a["x_t"]), cu(a["x_can"]))
        f_sw = sw.swap_module(cu(ref["f_can"]), cu(idv))
        f_rf = sw.refine_module(cu(ref["f_swap"]))
        # propagate the engine's own chain up to f_ref and decode with the ORACLE: error of the feature chain alone
        chain = sw.refine_module(sw.swap_module(sw.warping_module.warp(f_s, cu(a["x_t"]), cu(a["x_can"]))[0], cu(idv)))
        with torch.no_grad():
            seg_o = O.warping_forward(sds["warping_module"], chain.cpu(), kp_driving=a["x_t"], kp_source=a["x_can"])["out"]
            chain_psnr = O.psnr(O.spade_decoder(sds["spade_generator"], seg_o), ref["out"])
        print("frame %5d id %2d: full %.2f dB | G alone %.2f | W.fwd+G %.2f | engine F..R + oracle W.fwd/G %.2f | rel: F %.1e  warp %.1e occ %.1e  T %.1e  R %.1e  "
              "deform %.1e occ2 %.1e seg %.1e | kp motion %.3f" % (
                  fs, ids, full, g_only, wg, chain_psnr, rel(f_s, ref["f_s"]), rel(f_can, ref["f_can"]), rel(occ, ref["occ"]), rel(f_sw, ref["f_swap"]),
                  rel(f_rf, ref["f_ref"]), rel(wd["deformation"], ref["deformation"]), rel(wd["occlusion_map"], ref["occ2"]),
                  rel(sw.warping_module.warp_out(cu(ref["f_ref"]), cu(ref["occ2"])), ref["seg"]),
                  float((a["x_t"] - a["x_can"]).abs().mean())), flush=True)


if __name__ == "__main__":
    main()
