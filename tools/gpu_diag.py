"""GPU-box diagnostic: per-stage error table of the HIP engine vs the oracle + quick timings.
Usage (on the GPU box):  python tools/gpu_diag.py [--batch 4] [--no-oracle]
Writes a human-readable report to stdout; asserts nothing."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from canonswap_amd import synth  # noqa: E402
from canonswap_amd.can_swap_e2e import can_swapper  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    print("device:", torch.cuda.get_device_name(0), "cpu cores:", os.cpu_count())
    t = time.time(); sds_np = synth.make_state_dicts(0); sds = synth.to_torch(sds_np); print("weights gen %.1fs" % (time.time() - t))
    t = time.time(); sw = can_swapper(None, state_dicts=sds, max_batch=max(2, a.batch)); torch.cuda.synchronize()
    print("engine create+pack+upload %.1fs" % (time.time() - t))
    inp = synth.make_frame_inputs(2, seed=1000, size=256)
    idv = torch.from_numpy(synth.make_identity(7))
    args = {k: torch.from_numpy(v) for k, v in inp.items()}
    if not a.no_oracle:
        from oracle import canonswap_ref as O
        t = time.time(); ref = O.swap_frame(sds, args["img"], args["x_t"], args["x_can"], idv, debug=True)
        print("oracle 2 frames (with debug decodes) %.1fs" % (time.time() - t))
        cu = lambda x: x.cuda()
        rows = []
        def chk(name, fn):
            try:
                rows.append((name, fn()))
            except Exception as e:  # noqa: BLE001
                rows.append((name, "EXC " + repr(e)[:300]))
        chk("F  f_s rel", lambda: rel(sw.extract_feature_3d(cu(args["img"])), ref["f_s"]))
        def w1():
            f, occ = sw.warping_module.warp(cu(ref["f_s"]), cu(args["x_t"]), cu(args["x_can"]))
            return (rel(f, ref["f_can"]), rel(occ, ref["occ"]))
        chk("W.warp (f_can, occ) rel", w1)
        chk("T  f_swap rel", lambda: rel(sw.swap_module(cu(ref["f_can"]), cu(idv)), ref["f_swap"]))
        chk("R  f_ref rel", lambda: rel(sw.refine_module(cu(ref["f_swap"])), ref["f_ref"]))
        def w2():
            r = sw.warping_module(cu(ref["f_ref"]), kp_source=cu(args["x_can"]), kp_driving=cu(args["x_t"]))
            return (rel(r["deformation"], ref["deformation"]), rel(r["occlusion_map"], ref["occ2"]), rel(r["out"], ref["seg"]))
        chk("W.forward (deform, occ, seg) rel", w2)
        chk("warp_out seg rel", lambda: rel(sw.warping_module.warp_out(cu(ref["f_ref"]), cu(ref["occ2"])),
                                            O.warp_out(sds["warping_module"], ref["f_ref"], ref["occ2"])))
        chk("G  psnr(seg->img)", lambda: O.psnr(sw.spade_generator(feature=cu(ref["seg"])).cpu(), ref["out"]))
        def full():
            r = sw.swap_frames(cu(args["img"]), cu(args["x_t"]), cu(args["x_can"]), cu(idv), debug=True)
            return tuple(round(O.psnr(r[k].cpu(), ref[k]), 2) for k in ("out", "rec_can", "swap_can"))
        chk("swap_frames psnr (out, rec_can, swap_can)", full)
        for n, v in rows:
            print("  %-44s %s" % (n, v))
    # timing
    eng = sw.engine
    for B in sorted({1, a.batch}):
        inp = synth.make_frame_inputs(B, seed=3000, size=256)
        ga = [torch.from_numpy(inp[k]).cuda() for k in ("img", "x_t", "x_can")]
        out = torch.empty(B, 3, 512, 512, device="cuda")
        for _ in range(2):
            eng.swap_frames(*ga, idv.cuda(), out_f32=out)
        torch.cuda.synchronize(); t = time.time(); n = 5
        for _ in range(n):
            eng.swap_frames(*ga, idv.cuda(), out_f32=out)
        torch.cuda.synchronize(); dt = (time.time() - t) / n
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        os.environ["CANONSWAP_PROFILE_CSV"] = os.path.join(ROOT, "gpurun_out", "layers_b%d.csv" % B)
        eng.profile_begin(); eng.swap_frames(*ga, idv.cuda(), out_f32=out); p = eng.profile_end()
        del os.environ["CANONSWAP_PROFILE_CSV"]
        print("B=%d: %.2f ms/step  %.1f frames/s | conv %.2f ms (%d launches, %.1f TFLOP/s)  other %.2f ms (%d launches)  algo GFLOP/frame %.1f"
              % (B, dt * 1e3, B / dt, p["conv_ms"], p["conv_launches"], p["conv_flops"] / p["conv_ms"] / 1e9, p["other_ms"],
                 p["other_launches"], p["conv_flops"] / B / 1e9))


if __name__ == "__main__":
    main()
