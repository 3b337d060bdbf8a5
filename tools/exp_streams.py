"""Experiment (GPU box): S engines on S HIP streams, each stepping its own batch, vs one engine on one stream.
    python tools/exp_streams.py --streams 2 --batch 16 --steps 20"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from canonswap_amd import synth  # noqa: E402
from canonswap_amd.can_swap_e2e import can_swapper  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    sds = synth.to_torch(synth.make_state_dicts(0))
    sid = torch.from_numpy(synth.make_identity(7)).to(dev)
    S, B = a.streams, a.batch
    sws, sts, ins, outs = [], [], [], []
    for s in range(S):
        sw = can_swapper(type("Cfg", (), {"device_id": 0, "flag_force_cpu": False})(), state_dicts=sds, max_batch=B)
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            sw.engine.set_identity(sid)
        inp = synth.make_frame_inputs(B, seed=1000 + s, size=256)
        ins.append([torch.from_numpy(inp[k]).to(dev) for k in ("img", "x_t", "x_can")])
        outs.append(torch.empty(B, 512, 512, 3, dtype=torch.uint8, device=dev))
        sws.append(sw); sts.append(st)
    torch.cuda.synchronize()

    def step():
        for s in range(S):
            with torch.cuda.stream(sts[s]):
                sws[s].engine.swap_frames(*ins[s], want_f32=False, want_u8=True, out_u8=outs[s])

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"streams={S} batch/stream={B}: {dt / a.steps * 1e3:.2f} ms/step  {S * B * a.steps / dt:.1f} frames/s")


if __name__ == "__main__":
    main()
