"""conv_wide (persistent 256 x 256 tiles, 8 x 8 fragments per wave) against conv_halo's 128 x 256 tile: the same K order per output
element and the same epilogue macro, so the outputs must be equal bit for bit (torch.equal), and against plain PyTorch fp32 for one
case.  The T blend forms are the two the engine launches (adaptive_modulate.py:128-193, 337-349): conv1 (mask blend + ReLU, fp16 out)
and conv2 (mask blend + fp32 residual, fp32 stream out + fp16 copy with the next layer's affine / ReLU)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
CFG_HALO_128x256, CFG_WIDE = 17, 31


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _tblend_inputs(seed, N, H, W, Cin, Cout):
    from canonswap_amd import pack
    r = _rng(seed)
    x = np.maximum(r.standard_normal((N, 1, H, W, Cin)), 0).astype(np.float16)          # post-ReLU activations
    sc = 1.0 / np.sqrt(9 * Cin)
    w_std = (sc * r.standard_normal((Cout, Cin, 3, 3))).astype(np.float32)
    w_mod = (sc * r.standard_normal((Cout, Cin, 3, 3))).astype(np.float32)
    wp = torch.from_numpy(pack.pack_conv(pack.interleave16(w_std, w_mod), 2 * Cout)).to(DEV)
    bias = torch.from_numpy((0.1 * r.standard_normal(Cout)).astype(np.float32)).to(DEV)
    m4 = torch.zeros(N, H, W, 4, dtype=torch.float32, device=DEV)
    mask = r.uniform(0, 1, (N, H, W)).astype(np.float32)
    m4[..., 0] = torch.from_numpy(mask).to(DEV)
    res = torch.from_numpy(r.standard_normal((N, 1, H, W, Cout)).astype(np.float32)).to(DEV)
    s2 = torch.from_numpy(r.uniform(0.5, 1.5, Cout).astype(np.float32)).to(DEV)
    t2 = torch.from_numpy((0.2 * r.standard_normal(Cout)).astype(np.float32)).to(DEV)
    return dict(x=torch.from_numpy(x).to(DEV), wp=wp, bias=bias, m4=m4, res=res, s2=s2, t2=t2, w_std=w_std, w_mod=w_mod, mask=mask)


def _run(d, form, cfg, Cout):
    import hip_ops as ops
    N, _, H, W, _ = d["x"].shape
    if form == "conv1":
        out0 = torch.full((N, 1, H, W, Cout), -7.0, dtype=torch.float16, device=DEV)
        ops.conv(d["x"], d["wp"], 2 * Cout, Cout, (1, 3, 3), bias=d["bias"], pixscale=d["m4"], ps_stride=4, act0="relu", out0=out0, mode=1, cfg=cfg)
        torch.cuda.synchronize()
        return (out0,)
    out0 = torch.full((N, 1, H, W, Cout), -7.0, dtype=torch.float32, device=DEV)
    out1 = torch.full((N, 1, H, W, Cout), -7.0, dtype=torch.float16, device=DEV)
    ops.conv(d["x"], d["wp"], 2 * Cout, Cout, (1, 3, 3), bias=d["bias"], pixscale=d["m4"], ps_stride=4, res=d["res"], out0=out0,
             s2=d["s2"], t2=d["t2"], act1="relu", out1=out1, mode=1, cfg=cfg)
    torch.cuda.synchronize()
    return out0, out1


# N = 4: one item per workgroup; 6: uneven item counts; 16: four items per workgroup (cross-item prefetch); 32 x 32 maps: two tiles per row
@pytest.mark.parametrize("form", ["conv1", "conv2"])
@pytest.mark.parametrize("N,H,W", [(4, 64, 64), (6, 64, 64), (16, 64, 64), (5, 32, 32), (1, 16, 16)])
def test_wide_tblend_equals_halo(form, N, H, W):
    d = _tblend_inputs(100 + N, N, H, W, 512, 512)
    a = _run(d, form, CFG_HALO_128x256, 512)
    b = _run(d, form, CFG_WIDE, 512)
    for u, v in zip(a, b):
        assert torch.equal(u, v), (form, N, float((u.float() - v.float()).abs().max()))


def test_wide_tblend_smaller_channel_counts():
    """256 -> 2 x 128 (one channel block) and 128 -> 2 x 256 (two): the item order has other group sizes."""
    for Cin, Cout in ((256, 128), (128, 256)):
        d = _tblend_inputs(7, 8, 32, 32, Cin, Cout)
        for form in ("conv1", "conv2"):
            a = _run(d, form, CFG_HALO_128x256, Cout)
            b = _run(d, form, CFG_WIDE, Cout)
            for u, v in zip(a, b):
                assert torch.equal(u, v), (Cin, Cout, form)


def test_wide_tblend_vs_torch():
    """== AdaptiveSharedWeightConv2d's blend (adaptive_modulate.py:139-186) in fp32 on the fp16-rounded operands"""
    import hip_ops as ops
    N, H, W, C = 2, 32, 32, 256
    d = _tblend_inputs(3, N, H, W, C, C)
    out0, out1 = _run(d, "conv2", CFG_WIDE, C)
    xq = d["x"].float().cpu()[:, 0].permute(0, 3, 1, 2)
    q = lambda w: torch.from_numpy(w).half().float()
    m = torch.from_numpy(d["mask"]).unsqueeze(1)
    ref = m * (F.conv2d(xq, q(d["w_mod"]), None, padding=1) + d["bias"].cpu().view(1, -1, 1, 1)) + (1 - m) * F.conv2d(xq, q(d["w_std"]), None, padding=1)
    ref = ref + d["res"].cpu()[:, 0].permute(0, 3, 1, 2)
    got = out0.cpu()[:, 0].permute(0, 3, 1, 2)
    assert ops.rel_err(got, ref) < 1e-3
    ref1 = F.relu(ref * d["s2"].cpu().view(1, -1, 1, 1) + d["t2"].cpu().view(1, -1, 1, 1))
    assert ops.rel_err(out1.float().cpu()[:, 0].permute(0, 3, 1, 2), ref1) < 2e-3


def test_wide_refuses_what_it_does_not_cover():
    import hip_ops as ops
    d = _tblend_inputs(5, 1, 24, 24, 128, 128)          # 24 is not a multiple of the 16 x 16 tile
    with pytest.raises(RuntimeError):
        _run(d, "conv1", CFG_WIDE, 128)


def _std_inputs(seed, N, H, W, Cin, Cout):
    from canonswap_amd import pack
    r = _rng(seed)
    x = np.maximum(r.standard_normal((N, 1, H, W, Cin)), 0).astype(np.float16)
    w = (r.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    d = dict(x=torch.from_numpy(x).to(DEV), wp=torch.from_numpy(pack.pack_conv(w, Cout)).to(DEV),
             bias=torch.from_numpy((0.1 * r.standard_normal(Cout)).astype(np.float32)).to(DEV),
             res32=torch.from_numpy(r.standard_normal((N, 1, H, W, Cout)).astype(np.float32)).to(DEV),
             s2=torch.from_numpy(r.uniform(0.5, 1.5, Cout).astype(np.float32)).to(DEV),
             t2=torch.from_numpy((0.2 * r.standard_normal(Cout)).astype(np.float32)).to(DEV))
    d["res16"] = d["res32"].half()
    return d


@pytest.mark.parametrize("form", ["plain", "res32_two_outputs", "stat", "stat_res16"])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(8, 64, 64, 512, 512), (3, 32, 32, 256, 256), (2, 128, 128, 128, 256)])
def test_wide_std_forms_equal_halo(form, N, H, W, Cin, Cout):
    """R's 2-D pair (util.py:120-128: LeakyReLU'd fp16 out; fp32 residual stream out + fp16 copy) and G's 3x3 convs that also emit the
    next InstanceNorm's partial statistics (util.py:329-344), without / with the block's fp16 residual."""
    import hip_ops as ops
    d = _std_inputs(31 + N, N, H, W, Cin, Cout)
    outs = []
    for cfg in (CFG_HALO_128x256, CFG_WIDE):
        o0_16 = torch.full((N, 1, H, W, Cout), -7.0, dtype=torch.float16, device=DEV)
        o0_32 = torch.full((N, 1, H, W, Cout), -7.0, dtype=torch.float32, device=DEV)
        o1 = torch.full((N, 1, H, W, Cout), -7.0, dtype=torch.float16, device=DEV)
        so = torch.full((N * (W // 16) * (H // 8) * 2 * Cout * 2,), float('nan'), dtype=torch.float32, device=DEV)
        if form == "plain":
            ops.conv(d["x"], d["wp"], Cout, Cout, (1, 3, 3), bias=d["bias"], act0="lrelu", slope0=0.01, out0=o0_16, cfg=cfg)
            outs.append((o0_16,))
        elif form == "res32_two_outputs":
            ops.conv(d["x"], d["wp"], Cout, Cout, (1, 3, 3), bias=d["bias"], res=d["res32"], out0=o0_32, s2=d["s2"], t2=d["t2"], act1="lrelu", slope1=0.01,
                     out1=o1, cfg=cfg)
            outs.append((o0_32, o1))
        elif form == "stat":
            ops.conv(d["x"], d["wp"], Cout, Cout, (1, 3, 3), bias=d["bias"], out0=o0_16, stat_out=so, cfg=cfg)
            outs.append((o0_16, so))
        else:
            ops.conv(d["x"], d["wp"], Cout, Cout, (1, 3, 3), bias=d["bias"], res=d["res16"], out0=o0_16, stat_out=so, cfg=cfg)
            outs.append((o0_16, so))
        torch.cuda.synchronize()
    for u, v in zip(*outs):
        assert not bool(torch.isnan(v.float()).any()), form          # every partial-statistics slot was written
        assert torch.equal(u, v), (form, float((u.float() - v.float()).abs().max()))


@pytest.mark.parametrize("xshift", [0, 1])
@pytest.mark.parametrize("N,S,Cc", [(4, 64, 512), (2, 128, 128)])
def test_wide_spade_equals_halo(xshift, N, S, Cc):
    """gamma / beta convs + InstanceNorm modulation (util.py:295-302), the modulated tensor at the same or half the resolution"""
    import hip_ops as ops
    from canonswap_amd import pack
    r = _rng(50 + xshift + S)
    Sx = S >> xshift
    actv = torch.from_numpy(np.maximum(r.standard_normal((N, 1, S, S, 128)), 0).astype(np.float16)).to(DEV)
    x = torch.from_numpy((2 * r.standard_normal((N, Sx, Sx, Cc)) + 0.5).astype(np.float16)).to(DEV)
    wg = (0.02 * r.standard_normal((Cc, 128, 3, 3))).astype(np.float32); wb = (0.02 * r.standard_normal((Cc, 128, 3, 3))).astype(np.float32)
    bg = torch.from_numpy((0.1 * r.standard_normal(Cc)).astype(np.float32)).to(DEV)
    bb = torch.from_numpy((0.1 * r.standard_normal(Cc)).astype(np.float32)).to(DEV)
    stats = ops.chan_stats(x.reshape(N, Sx * Sx, Cc))
    wp = torch.from_numpy(pack.pack_conv(pack.interleave16(wg, wb), 2 * Cc)).to(DEV)
    outs = []
    for cfg in (CFG_HALO_128x256, CFG_WIDE):
        out = torch.full((N, 1, S, S, Cc), -7.0, dtype=torch.float16, device=DEV)
        ops.conv(actv, wp, 2 * Cc, Cc, (1, 3, 3), bias=bg, bias2=bb, res=x.unsqueeze(1), res_shift=xshift, stats=stats, act0="lrelu", slope0=0.2,
                 out0=out, mode=2, cfg=cfg)
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1]), float((outs[0].float() - outs[1].float()).abs().max())


@pytest.mark.parametrize("cfg", [CFG_HALO_128x256, CFG_WIDE])
def test_partial_statistics_match_a_numpy_emulation_of_the_reduction_order(cfg):
    """The per-(64 positions, channel) partial (sum, sum of squares) a conv epilogue emits for the next InstanceNorm: accumulated per lane over
    the 4 rows of a block in fp32 (sum of squares by fma), then added over the 16 columns in xor-butterfly order (conv_epilogue.h,
    ep_row_sum16: DPP row shifts whose lane-0 value equals that butterfly).  Emulated here step by step in numpy from the stored fp16
    output: equal bit for bit - the check that pins the reduction order independently of any kernel."""
    import hip_ops as ops
    N, H, W, C = 2, 32, 32, 256
    d = _std_inputs(91, N, H, W, C, C)
    nblk = (W // 16) * (H // 8) * 2
    o0 = torch.zeros(N, 1, H, W, C, dtype=torch.float16, device=DEV)
    so = torch.full((N * nblk * C * 2,), float("nan"), dtype=torch.float32, device=DEV)
    ops.conv(d["x"], d["wp"], C, C, (1, 3, 3), bias=d["bias"], out0=o0, stat_out=so, cfg=cfg)
    torch.cuda.synchronize()
    v = o0.cpu().numpy()[:, 0].astype(np.float32)                     # [N, H, W, C]
    got = so.cpu().numpy().reshape(N, nblk, C, 2)
    f32 = np.float32
    for n in range(N):
        for h8 in range(H // 8):
            for tw in range(W // 16):
                for sg in range(2):
                    blk = (h8 * (W // 16) + tw) * 2 + sg
                    t = v[n, h8 * 8 + sg * 4: h8 * 8 + sg * 4 + 4, tw * 16: tw * 16 + 16, :]         # [4 rows, 16 cols, C]
                    s = np.zeros((16, C), f32); q = np.zeros((16, C), f32)
                    for r in range(4):
                        s = (s + t[r]).astype(f32)
                        q = (t[r].astype(np.float64) * t[r].astype(np.float64) + q.astype(np.float64)).astype(f32)      # fma: one rounding
                    for o in (1, 2, 4, 8):
                        idx = np.arange(16) ^ o
                        s = (s + s[idx]).astype(f32); q = (q + q[idx]).astype(f32)
                    assert np.array_equal(got[n, blk, :, 0], s[0]) and np.array_equal(got[n, blk, :, 1], q[0]), (n, blk)
