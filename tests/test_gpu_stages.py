"""Stage-level and whole-frame parity of the HIP engine against the oracle and the golden vectors.

Floating-point tolerance (BASELINE.json north_star): PSNR >= 50 dB on the 3x512x512 image in [0,1] against the
fp32 CPU reference; intermediate feature tensors are checked by relative L2 error.  The engine computes with
fp16 operands / fp32 accumulation (fp32 residual streams), so bit-exactness is not expected.  The stage gates sit about 3x above
the errors measured on the GPU (tests/diag/stage_errors.py -> profiles/r03_t_stage_errors.txt), not at a generic tolerance.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PSNR_GATE = 50.0


@pytest.fixture(scope="module")
def swapper(state_dicts):
    from canonswap_amd.can_swap_e2e import can_swapper
    return can_swapper(None, state_dicts=state_dicts, max_batch=4)


@pytest.fixture(scope="module")
def case(state_dicts):
    """Two full-size frames through the oracle (about 15 s of CPU)."""
    from canonswap_amd import synth
    from oracle import canonswap_ref as O
    inp = synth.make_frame_inputs(2, seed=1000, size=256)
    idv = torch.from_numpy(synth.make_identity(7))
    args = {k: torch.from_numpy(v) for k, v in inp.items()}
    ref = O.swap_frame(state_dicts, args["img"], args["x_t"], args["x_can"], idv, debug=True)
    return args, idv, ref


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


def test_extract_feature_3d(swapper, case):
    args, _, ref = case
    f = swapper.extract_feature_3d(args["img"].cuda())
    assert f.shape == (2, 32, 16, 64, 64) and f.dtype == torch.float32
    assert _rel(f, ref["f_s"]) < 1e-3                       # measured 3.95e-4 (tests/diag/stage_errors.py, three seeds)


def test_warp(swapper, case):
    args, _, ref = case
    f_can, occ = swapper.warping_module.warp(ref["f_s"].cuda(), args["x_t"].cuda(), args["x_can"].cuda())
    assert _rel(occ, ref["occ"]) < 6e-4                     # measured 1.3e-4 .. 1.9e-4
    assert _rel(f_can, ref["f_can"]) < 3e-4                 # measured 0.7e-4 .. 1.0e-4


def test_swap_module(swapper, case):
    _, idv, ref = case
    out = swapper.swap_module(ref["f_can"].cuda(), idv.cuda())
    assert _rel(out, ref["f_swap"]) < 1e-3                  # measured 3.2e-4 .. 3.5e-4


def test_refine_module(swapper, case):
    _, _, ref = case
    out = swapper.refine_module(ref["f_swap"].cuda())
    assert _rel(out, ref["f_ref"]) < 6e-4                   # measured 2.0e-4 .. 2.1e-4 (split-precision GroupNorm convs)


def test_warp_decode(swapper, case):
    from oracle import canonswap_ref as O
    args, _, ref = case
    ret = swapper.warp_decode(ref["f_ref"].cuda(), args["x_can"].cuda(), args["x_t"].cuda())
    assert _rel(ret["deformation"], ref["deformation"]) < 3e-4     # measured 0.6e-4 .. 0.8e-4
    assert _rel(ret["occlusion_map"], ref["occ2"]) < 3e-3          # measured 4.3e-4 .. 9.0e-4
    assert O.psnr(ret["out"].cpu(), ref["out"]) >= 60.0                 # W.forward + G from oracle inputs: measured 64.9 .. 66.8 dB


def test_conv_decode(swapper, case, state_dicts):
    from oracle import canonswap_ref as O
    _, _, ref = case
    img = swapper.conv_decode(ref["f_can"].cuda(), ref["occ"].cuda())
    assert O.psnr(img.cpu(), ref["rec_can"]) >= PSNR_GATE
    seg = swapper.warping_module.warp_out(ref["f_ref"].cuda(), ref["occ2"].cuda())
    with torch.no_grad():
        assert _rel(seg, O.warp_out(state_dicts["warping_module"], ref["f_ref"], ref["occ2"])) < 1.5e-3      # measured 4.7e-4
        seg_noocc = swapper.warping_module.warp_out(ref["f_ref"].cuda())
        assert _rel(seg_noocc, O.warp_out(state_dicts["warping_module"], ref["f_ref"], None)) < 1.5e-3


def test_swap_frames_psnr_and_debug(swapper, case):
    """Whole loop body (can_swap_pipeline_e2e.py:242-263) for B=2 including the two debug decodes."""
    from oracle import canonswap_ref as O
    args, idv, ref = case
    r = swapper.swap_frames(args["img"].cuda(), args["x_t"].cuda(), args["x_can"].cuda(), idv.cuda(), debug=True, want_u8=True)
    for k in ("out", "rec_can", "swap_can"):
        p = O.psnr(r[k].cpu(), ref[k])
        assert p >= PSNR_GATE, (k, p)
    u8 = r["out_u8"].cpu().numpy()
    assert u8.shape == (2, 512, 512, 3) and u8.dtype == np.uint8
    assert np.array_equal(u8, O.parse_output(r["out"]))                     # device pack == reference parse_output
    du8 = u8.astype(np.float64) - O.parse_output(ref["out"]).astype(np.float64)    # both truncated to 8 bits
    assert 10 * np.log10(255.0 ** 2 / np.mean(du8 ** 2)) >= 48.0 and np.abs(du8).mean() < 0.6
    assert np.array_equal(swapper.parse_output(r["out"]), u8)


def test_golden_full_size_frame(swapper, golden):
    """The committed reference output (tests/golden, produced by the reference's own modules)."""
    from canonswap_amd import synth
    from oracle import canonswap_ref as O
    g = golden("frame_256_b1.npz")
    inp = synth.make_frame_inputs(1, seed=int(g["frame_seed"]), size=256)
    idv = torch.from_numpy(synth.make_identity(int(g["id_seed"])))
    r = swapper.swap_frames(*(torch.from_numpy(inp[k]).cuda() for k in ("img", "x_t", "x_can")), idv.cuda())
    assert O.psnr(r["out"].cpu(), torch.from_numpy(g["out_f16"].astype(np.float32))) >= PSNR_GATE


def test_batch_independence_and_determinism(swapper, case):
    """Frames are independent units (no temporal state): B=1 twice == B=2, bit for bit; reruns are bit-identical."""
    args, idv, _ = case
    a = swapper.swap_frames(args["img"].cuda(), args["x_t"].cuda(), args["x_can"].cuda(), idv.cuda())["out"].clone()
    b0 = swapper.swap_frames(args["img"][:1].cuda(), args["x_t"][:1].cuda(), args["x_can"][:1].cuda(), idv.cuda())["out"].clone()
    b1 = swapper.swap_frames(args["img"][1:].cuda(), args["x_t"][1:].cuda(), args["x_can"][1:].cuda(), idv.cuda())["out"].clone()
    assert torch.equal(a[0], b0[0]) and torch.equal(a[1], b1[0])       # no atomics, fixed reduction orders
    a2 = swapper.swap_frames(args["img"].cuda(), args["x_t"].cuda(), args["x_can"].cuda(), idv.cuda())["out"]
    assert torch.equal(a, a2)


def test_identity_changes_output(swapper, case):
    from canonswap_amd import synth
    args, idv, _ = case
    other = torch.from_numpy(synth.make_identity(8))
    a = swapper.swap_frames(args["img"][:1].cuda(), args["x_t"][:1].cuda(), args["x_can"][:1].cuda(), idv.cuda())["out"].clone()
    b = swapper.swap_frames(args["img"][:1].cuda(), args["x_t"][:1].cuda(), args["x_can"][:1].cuda(), other.cuda())["out"]
    assert (a - b).abs().mean() > 1e-4


def test_errors_are_loud(swapper):
    with pytest.raises(ValueError):
        swapper.extract_feature_3d(torch.zeros(1, 3, 128, 128).cuda())
    with pytest.raises(ValueError):
        swapper.extract_feature_3d(torch.zeros(9, 3, 256, 256).cuda())        # > max_batch


def test_stage_api_equals_fused_loop(swapper, case):
    """Calling the reference's stage methods one by one (with fp32 NCDHW tensors in between) gives the same image as
    the fused loop body: the only difference is where fp16 conv-input copies are derived, so agreement is to ~65 dB."""
    from oracle import canonswap_ref as O
    args, idv, _ = case
    img, x_t, x_can, sid = args["img"].cuda(), args["x_t"].cuda(), args["x_can"].cuda(), idv.cuda()
    f_s = swapper.extract_feature_3d(img)
    f_can, occ = swapper.warping_module.warp(f_s, x_t, x_can)
    f_swap = swapper.swap_module(f_can, sid)
    f_ref = swapper.refine_module(f_swap)
    out = swapper.warp_decode(f_ref, x_can, x_t)["out"]
    fused = swapper.swap_frames(img, x_t, x_can, sid)["out"]
    assert O.psnr(out.cpu(), fused.cpu()) > 60.0


def test_non_contiguous_and_cpu_inputs_are_accepted(swapper, case):
    args, idv, _ = case
    img = args["img"][:1]
    a = swapper.extract_feature_3d(img.cuda())
    b = swapper.extract_feature_3d(img)                                   # CPU tensor: moved to the engine's device
    c = swapper.extract_feature_3d(img.cuda().permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2))   # strided view
    assert torch.equal(a, b) and torch.equal(a, c)


def test_swap_without_identity_is_an_error(state_dicts):
    from canonswap_amd.can_swap_e2e import can_swapper
    sw = can_swapper(None, state_dicts=state_dicts, max_batch=1)
    with pytest.raises(RuntimeError, match="cs_set_identity"):
        sw.engine.swap(torch.zeros(1, 32, 16, 64, 64).cuda())
    sw.engine.close()


def test_missing_weights_are_an_error():
    from canonswap_amd import _lib
    from canonswap_amd.engine import Engine
    e = Engine(0, max_batch=1)
    with pytest.raises(RuntimeError, match="was not uploaded"):
        _lib.check(e.lib.cs_finalize_weights(e.h), "cs_finalize_weights")
    with pytest.raises(RuntimeError, match="cs_finalize_weights has not been called"):
        e.extract_feature_3d(torch.zeros(1, 3, 256, 256).cuda())
    e.close()


def test_animate_frames_v2i(swapper, case, state_dicts):
    """one feature volume + one source key-point set, B driving key-point sets
    (can_swap_pipeline_v2i.py:311-312) against the oracle's warp_decode and against the stage API."""
    from oracle import canonswap_ref as O
    args, _, ref = case
    f = ref["f_ref"][:1]
    ks, kd = args["x_can"][:1], args["x_t"]
    with torch.no_grad():
        seg = O.warping_forward(state_dicts["warping_module"], f.expand(2, -1, -1, -1, -1), kp_driving=kd, kp_source=ks.expand(2, -1, -1))["out"]
        want = O.spade_decoder(state_dicts["spade_generator"], seg)
    got = swapper.animate_frames(f.cuda(), ks.cuda(), kd.cuda(), want_u8=True)
    assert got["out"].shape == (2, 3, 512, 512) and got["out_u8"].shape == (2, 512, 512, 3)
    assert O.psnr(got["out"].cpu(), want) >= PSNR_GATE
    stage = swapper.warp_decode(f.expand(2, -1, -1, -1, -1).contiguous().cuda(), ks.expand(2, -1, -1).contiguous().cuda(), kd.cuda())["out"]
    assert O.psnr(got["out"].cpu(), stage.cpu()) > 60.0
    # the shared volume / key-point set is broadcast by a zero sample stride inside the kernels (round 6): the same bits as B copies of it
    rep = swapper.animate_frames(f.expand(2, -1, -1, -1, -1).contiguous().cuda(), ks.expand(2, -1, -1).contiguous().cuda(), kd.cuda(), want_u8=True)
    assert torch.equal(rep["out"], got["out"]) and torch.equal(rep["out_u8"], got["out_u8"])
    with pytest.raises(ValueError):
        swapper.animate_frames(ref["f_ref"][:2].repeat(2, 1, 1, 1, 1)[:3].cuda(), ks.cuda(), kd.cuda())


def test_max_batch_guard():
    from canonswap_amd.engine import Engine
    with pytest.raises(RuntimeError, match="max_batch"):
        Engine(0, max_batch=85)


def test_prepare_on_device_bit_exact(swapper):
    """A1 (can_swap_e2e.py:126-163): uint8 crops are uploaded as bytes and divided by 255 on the device - same fp32 values."""
    r = np.random.Generator(np.random.PCG64(5))
    img = r.integers(0, 256, size=(256, 256, 3), dtype=np.uint8)
    want = np.clip(img[np.newaxis].astype(np.float32) / 255., 0, 1).transpose(0, 3, 1, 2)
    got = swapper.prepare_source(img)
    assert got.shape == (1, 3, 256, 256) and got.dtype == torch.float32 and got.is_cuda
    assert np.array_equal(got.cpu().numpy(), want)
    vid = [r.integers(0, 256, size=(256, 256, 3), dtype=np.uint8) for _ in range(3)]
    v = swapper.prepare_videos(vid)
    assert v.shape == (3, 1, 3, 256, 256)
    wantv = np.clip(np.array(vid)[..., np.newaxis].astype(np.float32) / 255., 0, 1).transpose(0, 4, 3, 1, 2)
    assert np.array_equal(v.cpu().numpy(), wantv)
    f = swapper.prepare_source(img.astype(np.float32))          # float input keeps the host path
    assert np.array_equal(f.cpu().numpy(), want)


@pytest.mark.parametrize("frame_seed,id_seed,smooth", [(4321, 11, False), (777, 3, True), (90210, 29, True)])
def test_psnr_other_inputs(swapper, state_dicts, frame_seed, id_seed, smooth):
    """The 50 dB gate on further frames: other key-points / identities, and low-frequency (image-like) inputs."""
    from canonswap_amd import synth
    from oracle import canonswap_ref as O
    inp = synth.make_frame_inputs(1, seed=frame_seed, size=256)
    img = torch.from_numpy(synth.make_smooth_images(1, seed=frame_seed) if smooth else inp["img"])
    x_t, x_can = torch.from_numpy(inp["x_t"]), torch.from_numpy(inp["x_can"])
    idv = torch.from_numpy(synth.make_identity(id_seed))
    with torch.no_grad():
        ref = O.swap_frame(state_dicts, img, x_t, x_can, idv)
    out = swapper.swap_frames(img.cuda(), x_t.cuda(), x_can.cuda(), idv.cuda())["out"]
    assert O.psnr(out.cpu(), ref["out"]) >= PSNR_GATE


def test_empty_ragged_and_mistyped_inputs(swapper, case):
    """Empty batches, mismatched batch sizes and wrong ranks are refused before any launch; integer images are converted."""
    args, idv, _ = case
    img, x_t, x_can = args["img"].cuda(), args["x_t"].cuda(), args["x_can"].cuda()
    with pytest.raises(ValueError):
        swapper.swap_frames(img[:0], x_t[:0], x_can[:0], idv.cuda())                 # empty batch
    with pytest.raises(ValueError):
        swapper.swap_frames(img, x_t[:1].reshape(1, 63), x_can, idv.cuda())          # key-points not Bx21x3
    with pytest.raises(ValueError):
        swapper.swap_frames(img, x_t[:1], x_can, idv.cuda())                         # ragged: 2 frames, 1 key-point set
    with pytest.raises(ValueError):
        swapper.warp_decode(swapper.extract_feature_3d(img), x_can[:1], x_t)         # ragged stage call
    with pytest.raises(ValueError):
        swapper.warping_module.warp(torch.zeros(2, 32, 16, 64, 32).cuda(), x_t, x_can)  # truncated volume
    with pytest.raises(TypeError):
        swapper.extract_feature_3d(args["img"].numpy())                              # not a tensor
    with pytest.raises(ValueError):
        swapper.engine.set_identity(torch.randn(2, 512))                             # two different identities in one slot
    a = swapper.extract_feature_3d(img)
    b = swapper.extract_feature_3d(img.double())                                     # other float types are cast, same result
    assert torch.equal(a, b)


_SHORTCUT_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from canonswap_amd import synth
from canonswap_amd.can_swap_e2e import can_swapper
sd = synth.to_torch(synth.make_state_dicts(0))
sw = can_swapper(None, state_dicts=sd, max_batch=2)
r = np.random.default_rng(5)
f = torch.from_numpy(r.standard_normal((2, 32, 16, 64, 64)).astype(np.float32)).cuda()
occ = torch.from_numpy(r.random((2, 1, 64, 64)).astype(np.float32)).cuda()
img = sw.conv_decode(f, occ)
np.save({out!r}, img.cpu().numpy())
"""


def test_learned_shortcut_forms_agree(tmp_path):
    """G's learned shortcuts (util.py:329-344) in three forms: the reference's order (fused gamma/beta launch, then conv_s), conv_s(beta)
    composed into one conv at load time (three launches), and - for up_1 - conv_s inside the gamma conv's epilogue.  Same image to far inside
    the tolerance; each form is chosen once per process (environment), hence the subprocesses."""
    import os
    import subprocess
    import sys
    from oracle import canonswap_ref as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    imgs = []
    for i, env in enumerate(({"CANONSWAP_SHORTCUT_ALGEBRA": "0"}, {"CANONSWAP_SHORTCUT_FUSE": "0"}, {})):
        out = str(tmp_path / f"img{i}.npy")
        subprocess.run([sys.executable, "-c", _SHORTCUT_SCRIPT.format(root=root, out=out)], check=True, env={**os.environ, **env}, timeout=600)
        imgs.append(torch.from_numpy(np.load(out)))
    assert imgs[0].shape == (2, 3, 512, 512)
    for a in imgs[1:]:
        assert O.psnr(a, imgs[0]) > 65.0
    assert not torch.equal(imgs[1], imgs[2]) or True        # (another summation order of conv_s: equal bits are not required)
    assert O.psnr(imgs[2], imgs[1]) > 70.0
