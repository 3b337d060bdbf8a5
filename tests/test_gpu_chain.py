"""The whole device-side frame (canonswap_amd/chain.py; SURVEY 8f rows N1-N3 around the generator): uint8 crops ->
cs_prepare_crops -> cs_motion_extract + cs_motion_keypoints -> cs_swap_frames_ids -> cs_soft_erosion_frames -> cs_paste_back_batch -> uint8
frames.  Each batched step against the single-frame form it replaces (bit-equal) and against the oracle; the chain against the oracle's
composition of the reference's per-frame loop (can_swap_pipeline_e2e.py:223-283)."""
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


def _affine(k, Ho, Wo):
    th, sc = 0.1 * k - 0.15, 0.8 + 0.12 * k
    tx, ty = 0.3 * Wo - 40.5 * k, 0.1 * Ho + 33.25 * k
    return np.array([[sc * np.cos(th), -sc * np.sin(th), tx], [sc * np.sin(th), sc * np.cos(th), ty], [0, 0, 1]], np.float64)


def _masks(n, seed=5):
    r = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:512, 0:512].astype(np.float32)
    out = []
    for k in range(n):
        cx, cy, a, b = r.uniform(200, 312), r.uniform(200, 312), r.uniform(120, 200), r.uniform(150, 220)
        out.append((((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1).astype(np.uint8))
    return np.stack(out)


def test_keypoints_kernel_vs_oracle(swapper_m):
    """cs_motion_keypoints on raw head outputs the oracle also sees: x_t = s (kp R + exp) + t_xy, x_can = s kp, R = (Rz Ry Rx)^T."""
    from oracle import canonswap_ref as O
    r = np.random.Generator(np.random.PCG64(3))
    B = 4
    raw = r.normal(0, 1, size=(B, 328)).astype(np.float32)
    raw[:, :63] *= 0.3; raw[:, 63] = r.uniform(0.9, 1.3, B); raw[:, 262:265] *= 0.1; raw[:, 265:] *= 0.02
    raw[:, 64:262] *= 3.0                                   # peaked 66-bin pose logits
    t = torch.from_numpy(raw)
    info, o = {}, 0
    for k, n in O.M_HEADS:
        info[k] = t[:, o:o + n].clone(); o += n
    pose = [O.headpose_pred_to_degree(info[k]) for k in ("pitch", "yaw", "roll")]
    want_t = O.transform_keypoint(info)
    want_can = info["scale"][..., None] * info["kp"].reshape(B, 21, 3)
    want_R = O.get_rotation_matrix(*pose)
    x_t, x_can, R = swapper_m.engine.motion_keypoints(t.cuda(), want_rot=True)
    assert (x_t.cpu() - want_t).abs().max().item() <= 2e-6 * max(1.0, want_t.abs().max().item()) * 4
    assert torch.equal(x_can.cpu(), want_can)               # one fp32 product per element
    assert (R.cpu() - want_R).abs().max().item() <= 1e-6


def test_chain_keypoints_equal_the_torch_path(swapper_m):
    from canonswap_amd import synth
    from canonswap_amd.chain import FrameChain
    imgs = torch.from_numpy(synth.make_smooth_images(3, seed=2000, size=256)).cuda()
    info = swapper_m.get_kp_info(imgs)
    x_t = swapper_m.transform_keypoint(info)
    x_can = info["scale"][..., None] * info["kp"]
    got_t, got_can = FrameChain(swapper_m).keypoints(imgs)
    assert (got_t - x_t).abs().max().item() <= 2e-6 and (got_can - x_can).abs().max().item() <= 1e-7


def test_soft_erosion_frames_is_b_single_calls(swapper_m):
    """Per-frame maximum (the pipeline calls the module once per frame), uint8 labels == their float copy, and equal to the batch entry
    point run one sample at a time."""
    from canonswap_amd import tail
    e = swapper_m.engine
    m = torch.from_numpy(_masks(3)).cuda()
    m[1] = 0; m[1, 250:262, 250:262] = 1                    # a tiny mask: its own maximum is far below the others'
    se = swapper_m.soft_mask(21, 0.9, 3)
    got = tail.soft_erosion_frames(e, m, se.weight, 21, 0.9, 3)
    gotf = tail.soft_erosion_frames(e, m.float(), se.weight, 21, 0.9, 3)
    assert torch.equal(got, gotf)
    for k in range(3):
        one, _ = se(m[k:k + 1, None].float())
        assert torch.equal(got[k], one[0, 0]), k
    both, _ = se(m[:, None].float())                        # the module's own batch semantics: one maximum for the whole tensor
    assert not torch.equal(both[1, 0], got[1])


def test_soft_erosion_frames_vs_oracle(swapper_m):
    from canonswap_amd import tail
    from oracle import cv_ref as R
    m = _masks(2, seed=9)
    se = swapper_m.soft_mask(21, 0.9, 3)
    got = tail.soft_erosion_frames(swapper_m.engine, torch.from_numpy(m).cuda(), se.weight, 21, 0.9, 3).cpu().numpy()
    for k in range(2):
        want, hard = R.soft_erosion(torch.from_numpy(m[k:k + 1, None].astype(np.float32)), 21, 0.9, 3)
        want, hard = want.numpy()[0, 0], hard.numpy()[0, 0]
        flips = (got[k] >= 1.0) != hard
        assert flips.sum() <= 4
        assert np.abs(got[k] - want)[~flips].max() < 2e-6


@pytest.mark.parametrize("size", [(720, 1280), (301, 403)])
def test_paste_back_batch_equals_single_frames(swapper_m, size):
    """One launch for B frames (four pixels per thread, copies outside the crop) == the single-frame kernel, which is bit-exact against
    oracle/cv_ref.py (test_gpu_tail.py); the odd width takes the per-frame fallback."""
    from canonswap_amd import tail
    from oracle import cv_ref as R
    Ho, Wo = size
    e = swapper_m.engine
    r = np.random.Generator(np.random.PCG64(17))
    B = 3
    crops = r.integers(0, 256, size=(B, 512, 512, 3), dtype=np.uint8)
    ori = r.integers(0, 256, size=(B, Ho, Wo, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:512, 0:512].astype(np.float32)
    masks = np.stack([np.clip(1.2 - np.hypot(xx - 256, yy - 256) / (160 + 30 * k), 0, 1) for k in range(B)]).astype(np.float32)
    Ms = np.stack([_affine(k, Ho, Wo) for k in range(B)])
    Ms[2, 0, 2] = -150.5                                   # partly outside the frame
    got = tail.paste_back_batch(e, crops, masks, Ms, ori).cpu().numpy()
    for k in range(B):
        one = tail.paste_back_fused(e, crops[k], masks[k], Ms[k], ori[k]).cpu().numpy()
        assert np.array_equal(got[k], one), k
    mo = R.prepare_paste_back(np.stack([masks[0]] * 3, -1), Ms[0], (Wo, Ho))
    assert np.array_equal(got[0], R.paste_back(crops[0], Ms[0], ori[0], mo))


def test_chain_vs_oracle_loop(swapper_m, sds_m):
    """The chain for B = 2 frames against the oracle's composition of the reference's per-frame loop, and bit-equal to its own stages
    called one by one."""
    from canonswap_amd import synth, tail
    from canonswap_amd.chain import FrameChain
    from oracle import canonswap_ref as O
    from oracle import cv_ref as R
    B, Ho, Wo = 2, 540, 960
    smooth = synth.make_smooth_images(B, seed=2100, size=512)                       # (B,3,512,512) in [0,1]
    crops = np.ascontiguousarray((smooth.transpose(0, 2, 3, 1) * 255).astype(np.uint8))
    masks = _masks(B, seed=21)
    r = np.random.Generator(np.random.PCG64(23))
    ori = r.integers(0, 256, size=(B, Ho, Wo, 3), dtype=np.uint8)
    Ms = np.stack([_affine(k, Ho, Wo) * np.array([[0.5], [0.5], [1]]) + np.array([[0, 0, 100.], [0, 0, 20.], [0, 0, 0]]) for k in range(B)])
    idv = torch.from_numpy(synth.make_identity(7))
    chain = FrameChain(swapper_m)
    res = chain(torch.from_numpy(crops).cuda(), torch.from_numpy(masks).cuda(), Ms, torch.from_numpy(ori).cuda(), idv.cuda(), keep=True)
    got = res["frames"].cpu().numpy()
    # ---- the stages one by one: the same bits
    e = swapper_m.engine
    I = tail.prepare_crops(e, crops)
    assert torch.equal(I, res["I"])
    gen = e.swap_frames(I, res["x_t"], res["x_can"], idv.cuda(), want_f32=False, want_u8=True)["out_u8"]
    assert torch.equal(gen, res["crops_out"])
    for k in range(B):
        soft, _ = chain.se(torch.from_numpy(masks[k:k + 1, None]).cuda().float())
        one = tail.paste_back_fused(e, gen[k], soft[0, 0], Ms[k], ori[k])
        assert np.array_equal(one.cpu().numpy(), got[k]), k
    # ---- the oracle's loop
    worst = 1e9
    for k in range(B):
        with torch.no_grad():
            Ik = O.prepare_source(R.resize_area_2x_u8(crops[k]))
            info = O.get_kp_info(sds_m["motion_extractor"], Ik)
            x_t = O.transform_keypoint(info)
            x_can = info["scale"][..., None] * info["kp"]
            assert (res["x_t"][k].cpu() - x_t[0]).abs().max().item() <= 1e-4
            out = O.swap_frame(sds_m, Ik, x_t, x_can, idv)["out"]
        crop_o = O.parse_output(out)[0]
        soft, _ = R.soft_erosion(torch.from_numpy(masks[k:k + 1, None].astype(np.float32)), 21, 0.9, 3)
        soft = soft.numpy()[0, 0]
        mo = R.prepare_paste_back(np.stack([soft] * 3, -1), Ms[k], (Wo, Ho))
        want = R.paste_back(crop_o, Ms[k], ori[k], mo)
        region = mo[..., 0] > 0
        assert np.array_equal(got[k][~region], want[~region])                       # untouched pixels are the original frame
        d = got[k][region].astype(np.float64) - want[region].astype(np.float64)
        p = 10 * np.log10(255.0 ** 2 / max((d ** 2).mean(), 1e-12))
        print(f"frame {k}: pasted region {int(region.sum())} px, PSNR {p:.2f} dB, mean |diff| {np.abs(d).mean():.3f} LSB")
        worst = min(worst, p)
        assert np.abs(d).mean() < 0.6
    assert worst >= 48.0                                                           # uint8 frames: the gate of the generator's u8 output


def test_prefetched_stage_a_gives_the_same_frames(swapper_m):
    """FrameChain.prefetch: staging + M + key-points + soft masks of the next batch on a side stream (double-buffered) beside the current batch's
    generator - the same bytes as the in-line chain, batch after batch."""
    from canonswap_amd import synth
    from canonswap_amd.chain import FrameChain
    B, Ho, Wo = 2, 360, 640
    r = np.random.Generator(np.random.PCG64(29))
    idv = torch.from_numpy(synth.make_identity(7)).cuda()
    batches = []
    for k in range(3):
        smooth = synth.make_smooth_images(B, seed=2200 + k, size=512)
        crops = torch.from_numpy(np.ascontiguousarray((smooth.transpose(0, 2, 3, 1) * 255).astype(np.uint8))).cuda()
        masks = torch.from_numpy(_masks(B, seed=31 + k)).cuda()
        ori = torch.from_numpy(r.integers(0, 256, size=(B, Ho, Wo, 3), dtype=np.uint8)).cuda()
        Ms = np.stack([_affine(j, Ho, Wo) * np.array([[0.4], [0.4], [1]]) + np.array([[0, 0, 60.], [0, 0, 10.], [0, 0, 0]]) for j in range(B)])
        batches.append((crops, masks, Ms, ori))
    chain = FrameChain(swapper_m)
    want = [chain(c, m, M, o, idv)["frames"].clone() for c, m, M, o in batches]
    chain.prefetch(batches[0][0], batches[0][1])
    got = []
    for k, (c, m, M, o) in enumerate(batches):
        if k + 1 < len(batches):
            chain.prefetch(batches[k + 1][0], batches[k + 1][1])
        got.append(chain(c, m, M, o, idv)["frames"].clone())
    torch.cuda.synchronize()
    for k in range(3):
        assert torch.equal(got[k], want[k]), k
    assert not torch.equal(want[0], want[1])
    with pytest.raises(RuntimeError):
        chain.prefetch(batches[0][0], batches[0][1]); chain.prefetch(batches[1][0], batches[1][1]); chain.prefetch(batches[2][0], batches[2][1])
    chain.drop_prefetches()
