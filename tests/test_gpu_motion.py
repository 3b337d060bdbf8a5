"""Motion extractor M on the HIP engine against the oracle and the reference's own vectors.

M's GEMMs run in split precision (hi/lo fp16 operand pairs, fp32 accumulate; csrc/motion.hip), LayerNorm / GRN / GELU and
the residual stream in fp32: plain fp16 operands gave 1e-3 on the key-points and only 41 dB on the generated frame, the
split form measures 2.5e-6.  Gates: |d kp|, |d exp|, |d t|, |d scale| <= 1e-4, head-pose <= 0.01 degree, transformed
key-points <= 1e-4, and >= 50 dB on the generated frame when the key-points come from the HIP M instead of the oracle's.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sds_m():
    from canonswap_amd import synth
    return synth.to_torch(synth.make_state_dicts(0, modules=synth.MODULES + ("motion_extractor",)))


@pytest.fixture(scope="module")
def swapper_m(sds_m):
    from canonswap_amd.can_swap_e2e import can_swapper
    return can_swapper(None, state_dicts=sds_m, max_batch=4)


@pytest.fixture(scope="module")
def imgs():
    from canonswap_amd import synth
    return torch.from_numpy(synth.make_smooth_images(3, seed=2000, size=256))


def _chk(out, ref):
    for k in ("kp", "exp", "t", "scale"):
        d = (out[k].float().cpu().reshape(-1) - ref[k].reshape(-1)).abs().max().item()
        assert d <= 1e-4, (k, d)


def test_raw_heads_vs_oracle(swapper_m, sds_m, imgs):
    from oracle import canonswap_ref as O
    with torch.no_grad():
        ref = O.motion_extractor(sds_m["motion_extractor"], imgs)
    out = swapper_m.motion_extractor(imgs.cuda())
    assert set(out) == set(ref)
    for k, n in O.M_HEADS:
        assert out[k].shape == (3, n) and out[k].dtype == torch.float32
    _chk(out, ref)
    for k in ("pitch", "yaw", "roll"):     # 66-bin logits -> expected degrees (camera.py:14-28)
        dd = (O.headpose_pred_to_degree(out[k].cpu()) - O.headpose_pred_to_degree(ref[k])).abs().max().item()
        assert dd <= 0.01, (k, dd)


def test_raw_heads_vs_reference_vectors(swapper_m, golden, imgs):
    g = golden("motion_b3.npz")            # written by tools/make_golden.py from the reference's MotionExtractor
    out = swapper_m.motion_extractor(imgs.cuda())
    _chk(out, {k: torch.from_numpy(g[k]) for k in ("kp", "exp", "t", "scale")})


def test_get_kp_info_and_transform(swapper_m, sds_m, imgs):
    from oracle import canonswap_ref as O
    with torch.no_grad():
        ref = O.get_kp_info(sds_m["motion_extractor"], imgs)
        xr = O.transform_keypoint(ref)
    info = swapper_m.get_kp_info(imgs.cuda())
    assert info["kp"].shape == (3, 21, 3) and info["exp"].shape == (3, 21, 3) and info["pitch"].shape == (3, 1)
    x = swapper_m.transform_keypoint(info)
    assert x.shape == (3, 21, 3)
    assert (x.cpu() - xr).abs().max().item() <= 1e-4
    raw = swapper_m.get_kp_info(imgs.cuda(), flag_refine_info=False)
    assert raw["pitch"].shape == (3, 66) and raw["kp"].shape == (3, 63)


def test_batch_independence(swapper_m, imgs):
    a = swapper_m.motion_extractor(imgs.cuda())
    b = swapper_m.motion_extractor(imgs[1:2].cuda())
    for k in a:
        assert torch.equal(a[k][1:2], b[k]), k


def test_frame_with_hip_keypoints(swapper_m, sds_m, imgs):
    """M feeding the generator: x_t from the driving frame, x_can = scale * kp (can_swap_pipeline_e2e.py:236-241)."""
    from canonswap_amd import synth
    from oracle import canonswap_ref as O
    idv = torch.from_numpy(synth.make_identity(7))
    frame = imgs[:1]
    info = swapper_m.get_kp_info(frame.cuda())
    x_t = swapper_m.transform_keypoint(info)
    x_can = info["scale"][..., None] * info["kp"]
    with torch.no_grad():
        rinfo = O.get_kp_info(sds_m["motion_extractor"], frame)
        rx_t, rx_can = O.transform_keypoint(rinfo), rinfo["scale"][..., None] * rinfo["kp"]
        ref = O.swap_frame(sds_m, frame, rx_t, rx_can, idv)
    out = swapper_m.swap_frames(frame.cuda(), x_t, x_can, idv.cuda())["out"]
    assert O.psnr(out.cpu(), ref["out"]) >= 50.0


def test_missing_motion_weights(state_dicts):
    from canonswap_amd.can_swap_e2e import can_swapper
    sw = can_swapper(None, state_dicts=state_dicts, max_batch=1)
    with pytest.raises(RuntimeError):
        sw.get_kp_info(torch.zeros(1, 3, 256, 256, device="cuda"))


def test_pose_and_source_helpers(swapper_m, sds_m, imgs):
    """get_pose_dct / get_fs_and_kp_info / calc_ratio (can_swap_e2e.py:201-226, 324-331) on the engine."""
    from oracle import canonswap_ref as O
    raw = swapper_m.get_kp_info(imgs[:1].cuda(), flag_refine_info=False)
    pose = swapper_m.get_pose_dct(raw)
    with torch.no_grad():
        ref = O.motion_extractor(sds_m["motion_extractor"], imgs[:1])
    for k in ("pitch", "yaw", "roll"):
        assert abs(pose[k] - O.headpose_pred_to_degree(ref[k]).item()) <= 0.01
    s_info, s_rot, f_s, d_info, d_rot = swapper_m.get_fs_and_kp_info(imgs[:1].cuda(), imgs[1:2].cuda())
    assert f_s.shape == (1, 32, 16, 64, 64) and s_rot.shape == (1, 3, 3) and d_info["kp"].shape == (1, 21, 3)
    lmk = [np.random.default_rng(0).uniform(0, 256, size=(106, 2)).astype(np.float32) for _ in range(2)]
    eyes, lips = swapper_m.calc_ratio(lmk)
    assert len(eyes) == 2 and eyes[0].shape == (1, 2) and lips[0].shape == (1, 1)
    assert swapper_m.calc_combined_eye_ratio(eyes[0], lmk[1]).shape == (1, 3)
    assert swapper_m.calc_combined_lip_ratio(lips[0], lmk[1]).shape == (1, 2)
