"""Single-frame latency mode (BASELINE configs[1]): forms that only pay when a launch cannot fill the chip and that add an output
element's products in another order than the batched path - conv_lat (the K loop of the 512-channel 3x3 convs split over twelve waves of a workgroup),
cross-workgroup split-K for the deep hourglass levels, 2-row statistics blocks in R's volume convs.  Same tolerance as the batched path (PSNR >= 50 dB
vs the oracle), deterministic, and close to - not bit-identical with - the batched path, which is why it is a mode."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_latency_mode_frame(state_dicts):
    from canonswap_amd import synth
    from canonswap_amd.can_swap_e2e import can_swapper
    from oracle import canonswap_ref as O
    inp = synth.make_frame_inputs(1, seed=1000, size=256)
    args = [torch.from_numpy(inp[k]) for k in ("img", "x_t", "x_can")]
    idv = torch.from_numpy(synth.make_identity(7))
    with torch.no_grad():
        ref = O.swap_frame(state_dicts, *args, idv)
    lat = can_swapper(None, state_dicts=state_dicts, max_batch=1, latency_mode=True)
    std = can_swapper(None, state_dicts=state_dicts, max_batch=1)
    try:
        g = [a.cuda() for a in args]
        a1 = lat.swap_frames(*g, idv.cuda(), want_u8=True)
        a2 = lat.swap_frames(*g, idv.cuda(), want_u8=True)
        b = std.swap_frames(*g, idv.cuda(), want_u8=True)
        assert torch.equal(a1["out"], a2["out"])                                  # deterministic (fixed split order)
        p = O.psnr(a1["out"].cpu(), ref["out"])
        assert p >= 50.0, p
        assert O.psnr(a1["out"].cpu(), b["out"].cpu()) > 60.0                     # the two modes agree far inside the tolerance
        wd_l = lat.warp_decode(ref["f_ref"].cuda(), args[2].cuda(), args[1].cuda())
        wd_s = std.warp_decode(ref["f_ref"].cuda(), args[2].cuda(), args[1].cuda())
        for k in ("deformation", "occlusion_map"):                               # the split layers sit in the dense-motion hourglass
            d = float((wd_l[k] - wd_s[k]).abs().max())
            assert d < 2e-3, (k, d)
        assert float((wd_l["deformation"] - ref["deformation"].cuda()).norm() / ref["deformation"].norm()) < 2e-3
    finally:
        lat.engine.close(); std.engine.close()


@pytest.mark.parametrize("B,lat", [(2, False), (1, True)])
def test_grouped_phase_launch_equals_four_launches(state_dicts, B, lat):
    """The four output phases of the hourglass' two last up-sampling convs (util.py:142-147 on the source grid) run as ONE conv_halo launch
    (ConvParams::nphase, blockIdx.z = phase: weights, leading padding and output offset per phase); CANONSWAP_PHASE_GROUP=0 (subprocess: the knob
    is read once per process) launches them one by one - in latency mode as split-K convs with their finishing launches.  Batched path: the
    same bits.  Latency mode: the grouped launch runs the whole K loop in one workgroup where the four launches split it: agreement to 2e-3 on the
    deformation / occlusion map and beyond 60 dB on the image, as between the two modes."""
    import os
    import subprocess
    import sys
    from canonswap_amd import synth
    from canonswap_amd.can_swap_e2e import can_swapper
    from oracle import canonswap_ref as O
    inp = synth.make_frame_inputs(B, seed=1234, size=256)
    args = [torch.from_numpy(inp[k]) for k in ("img", "x_t", "x_can")]
    idv = torch.from_numpy(synth.make_identity(7))
    with torch.no_grad():
        ref = O.swap_frame(state_dicts, *[a[:1] for a in args], idv)
    f_ref = ref["f_ref"].repeat(B, 1, 1, 1, 1)
    sw = can_swapper(None, state_dicts=state_dicts, max_batch=B, latency_mode=lat)
    try:
        wd = sw.warp_decode(f_ref.cuda(), args[2].cuda(), args[1].cuda())
        got = {k: wd[k].cpu() for k in ("deformation", "occlusion_map", "out")}
    finally:
        sw.engine.close()
    torch.save({"f": f_ref, "kc": args[2], "kt": args[1]}, "/tmp/pg_in.pt")
    code = ("import torch, sys; sys.path.insert(0, %r); from canonswap_amd import synth; from canonswap_amd.can_swap_e2e import can_swapper;"
            "sd = synth.to_torch(synth.make_state_dicts(0)); sw = can_swapper(None, state_dicts=sd, max_batch=%d, latency_mode=%r);"
            "d = torch.load('/tmp/pg_in.pt'); wd = sw.warp_decode(d['f'].cuda(), d['kc'].cuda(), d['kt'].cuda());"
            "torch.save({k: wd[k].cpu() for k in ('deformation', 'occlusion_map', 'out')}, sys.argv[1])"
            ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), B, lat)
    env = dict(os.environ, CANONSWAP_PHASE_GROUP="0")
    r = subprocess.run([sys.executable, "-c", code, "/tmp/pg_off.pt"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = torch.load("/tmp/pg_off.pt")
    for k in got:
        if lat and k == "out":
            assert O.psnr(got[k], want[k]) > 60.0
        elif lat:
            assert float((got[k] - want[k]).abs().max()) < 2e-3, k
        else:
            assert torch.equal(got[k], want[k]), (k, float((got[k] - want[k]).abs().max()))
