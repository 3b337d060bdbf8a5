"""Operator-level parity of the HIP kernels (through the C ABI) against plain PyTorch fp32 CPU ops."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _randn(r, *shape, scale=1.0):
    return torch.from_numpy((scale * r.standard_normal(shape)).astype(np.float32))


def _to_cl(x):
    """NC(D)HW fp32 -> [N, D, H, W, C] fp16 channels-last (contiguous)."""
    if x.dim() == 4:
        x = x.unsqueeze(2)
    return x.permute(0, 2, 3, 4, 1).contiguous().half()


def _from_cl(y):
    """[N, D, H, W, C] -> NCDHW fp32 cpu."""
    return y.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def _ref_conv(x, w, b, pad):
    return F.conv3d(x.half().float(), w.half().float(), b, padding=pad)


CASES = [
    # name, N, Cin, Cout, D, H, W, k, cfg
    ("2d_3x3_128x128", 2, 64, 128, 1, 32, 32, (1, 3, 3), -1),
    ("2d_3x3_batch3", 3, 96, 256, 1, 16, 16, (1, 3, 3), -1),
    ("2d_1x1_n64", 1, 256, 64, 1, 32, 32, (1, 1, 1), -1),
    ("3d_3x3x3_n32", 1, 32, 32, 16, 16, 16, (3, 3, 3), -1),
    ("3d_small_spatial", 3, 64, 128, 16, 2, 2, (3, 3, 3), -1),
    ("3d_4x4", 2, 128, 64, 16, 4, 4, (3, 3, 3), -1),
    ("2d_n16", 1, 64, 16, 1, 32, 32, (1, 3, 3), -1),
    ("3d_7x7x7", 1, 48, 32, 8, 8, 8, (7, 7, 7), -1),
]


KERNELS = ["igemm", "halo"]


def _cfg(kern, igemm_cfg=-1, halo_cfg=-2):
    return igemm_cfg if kern == "igemm" else halo_cfg


@pytest.mark.parametrize("kern", KERNELS)
@pytest.mark.parametrize("name,N,Cin,Cout,D,H,W,k,cfg", CASES)
def test_conv_std(name, N, Cin, Cout, D, H, W, k, cfg, kern):
    import hip_ops as ops
    r = _rng(zlib.crc32(name.encode()) % 1000)
    x = _randn(r, N, Cin, D, H, W)
    w = _randn(r, Cout, Cin, *k, scale=1.0 / np.sqrt(Cin * np.prod(k)))
    b = _randn(r, Cout, scale=0.1)
    ref = F.relu(_ref_conv(x, w, b, tuple(kk // 2 for kk in k)))
    cout_pad = Cout
    xd = _to_cl(x).to(DEV)
    wp = ops.packed_weight(w, cout_pad, DEV)
    out = torch.zeros(N, D, H, W, Cout, dtype=torch.float32, device=DEV)
    ops.conv(xd, wp, cout_pad, Cout, k, bias=b.to(DEV), act0="relu", out0=out, cfg=_cfg(kern))
    torch.cuda.synchronize()
    err = ops.rel_err(_from_cl(out), ref)
    assert err < 2e-3, (name, kern, err)


@pytest.mark.parametrize("kern", KERNELS)
def test_conv_channel_slice_pad_and_upshift(kern):
    """Cin = 112 read at channel offset 32 of a 144-wide buffer (dense-motion level 0) with nearest x2 up-sampling
    folded into addressing (UpBlock3d, util.py:142-147) and output into a wider concat buffer."""
    import hip_ops as ops
    r = _rng(11)
    N, D, Hs, Ws = 2, 4, 8, 8
    x = _randn(r, N, 110, D, Hs, Ws)
    w = _randn(r, 64, 110, 3, 3, 3, scale=0.03)
    b = _randn(r, 64, scale=0.1)
    xu = x.repeat_interleave(2, 3).repeat_interleave(2, 4)
    ref = F.relu(_ref_conv(xu, w, b, 1))
    buf = torch.zeros(N, D, Hs, Ws, 144, dtype=torch.float16, device=DEV)
    buf[..., 32:142] = _to_cl(x).to(DEV)
    buf[..., :32] = 7.0                         # neighbouring channels must not leak in
    wp = ops.packed_weight(w, 64, DEV)
    obuf = torch.full((N, D, 2 * Hs, 2 * Ws, 96), -5.0, dtype=torch.float16, device=DEV)
    ops.conv(buf[..., 32:], wp, 64, 64, (3, 3, 3), cin=112, bias=b.to(DEV), act0="relu", out0=obuf[..., :64], up_shift=1,
             out_dims=(N, D, 2 * Hs, 2 * Ws), cfg=_cfg(kern))
    torch.cuda.synchronize()
    assert ops.rel_err(_from_cl(obuf[..., :64]), ref) < 3e-3
    assert bool((obuf[..., 64:] == -5.0).all())


@pytest.mark.parametrize("S,tile,Cin,Cout", [(8, (8, 8), 128, 256), (16, (8, 8), 64, 128), (4, (4, 4), 128, 512)])
def test_upblock_phase_conv_on_wide_tiles(S, tile, Cin, Cout):
    """UpBlock3d (util.py:142-147: nearest (1,2,2) up-sampling, 3x3x3 conv, folded BN, ReLU) per output phase on the SOURCE grid - a 3x2x2 conv
    with the phase's summed weights (pack.upsampled_conv3d_phases) - on the 128x128 tiles the hourglass' up-blocks 0 - 2 run on since round 6
    (static shapes 13 / 17: 8x8x2 and 4x4x8 positions): every phase, written at its offset into the full-resolution output, against the 27-tap
    conv on the up-sampled input.  (The engine launches the four phases as one grouped launch: tests/test_gpu_latency.py holds grouped == single.)"""
    import hip_ops as ops
    from canonswap_amd import pack
    r = _rng(40 + S)
    N, D = 2, 16
    x = _randn(r, N, Cin, D, S, S)
    w = _randn(r, Cout, Cin, 3, 3, 3, scale=0.03)
    b = _randn(r, Cout, scale=0.1)
    xu = x.repeat_interleave(2, 3).repeat_interleave(2, 4)
    ref = F.relu(_ref_conv(xu, w, b, 1))
    xd = _to_cl(x).to(DEV)
    out = torch.zeros(N, D, 2 * S, 2 * S, Cout, dtype=torch.float16, device=DEV)
    for (a, bb), (ph, pw, wab) in pack.upsampled_conv3d_phases(w.numpy()).items():
        if (ph, pw) != (1, 1):
            continue      # cs_op_conv pads KH / 2 = KW / 2 = 1 on the leading side: the (a, b) = (0, 0) phase; the others differ in the padding side only
        wp = torch.from_numpy(pack.pack_conv(wab, Cout)).to(DEV)
        ops.conv(xd, wp, Cout, Cout, (3, 2, 2), bias=b.to(DEV), act0="relu", out0=out[:, :, a::2, bb::2], cfg=10, tile=tile)
        torch.cuda.synchronize()
        got = _from_cl(out[:, :, a::2, bb::2])
        # the phase's weights are sums of two or four fp32 taps rounded to fp16 once: compare against the 27-tap result at a tolerance that covers it
        assert ops.rel_err(got, ref[:, :, :, a::2, bb::2]) < 4e-3, (S, a, bb)


@pytest.mark.parametrize("kern", KERNELS)
def test_conv_hwdc_residual_dual_output(kern):
    """3x3x3 conv on the [H][W][D][C] feature-volume layout, fp32 residual stream, second pre-activated fp16 output
    (ResBlock3d, util.py:94-102)."""
    import hip_ops as ops
    r = _rng(12)
    N, Cc, D, H, W = 2, 32, 16, 8, 8
    x = _randn(r, N, Cc, D, H, W)
    res = _randn(r, N, Cc, D, H, W)
    w = _randn(r, Cc, Cc, 3, 3, 3, scale=0.04)
    b = _randn(r, Cc, scale=0.1)
    s2 = torch.from_numpy(r.uniform(0.5, 1.5, Cc).astype(np.float32)); t2 = _randn(r, Cc, scale=0.2)
    y = _ref_conv(x, w, b, 1) + res
    y2 = F.relu(y * s2.view(1, -1, 1, 1, 1) + t2.view(1, -1, 1, 1, 1))
    hwdc = lambda t: t.permute(0, 3, 4, 2, 1).contiguous()           # NCDHW -> N H W D C
    xd = hwdc(x).half().to(DEV)
    resd = hwdc(res).to(DEV)
    view = lambda t: t.permute(0, 3, 1, 2, 4)                        # N H W D C -> logical [N, D, H, W, C] view
    out0 = torch.zeros(N, H, W, D, Cc, dtype=torch.float32, device=DEV)
    out1 = torch.zeros(N, H, W, D, Cc, dtype=torch.float16, device=DEV)
    wp = ops.packed_weight(w, 32, DEV)
    ops.conv(view(xd), wp, 32, 32, (3, 3, 3), bias=b.to(DEV), res=view(resd), out0=view(out0), s2=s2.to(DEV), t2=t2.to(DEV),
             act1="relu", out1=view(out1), tile=(4, 4), cfg=_cfg(kern, -1, 12))
    torch.cuda.synchronize()
    back = lambda t: t.float().cpu().permute(0, 4, 3, 1, 2)          # N H W D C -> N C D H W
    assert ops.rel_err(back(out0), y) < 2e-3
    assert ops.rel_err(back(out1), y2) < 3e-3


@pytest.mark.parametrize("kern", KERNELS + ["halo256"])
def test_conv_tblend(kern):
    """Fused [W ; w_mod] conv + blend epilogue == AdaptiveSharedWeightConv2d (adaptive_modulate.py:139-186)."""
    import hip_ops as ops
    from canonswap_amd import pack
    r = _rng(13)
    N, Cc, H, W = 2, 128, 16, 16
    x = _randn(r, N, Cc, H, W)
    w_std = _randn(r, Cc, Cc, 3, 3, scale=0.03); w_mod = _randn(r, Cc, Cc, 3, 3, scale=0.03)
    bias = _randn(r, Cc, scale=0.1)
    mask = torch.from_numpy(r.uniform(0, 1, (N, 1, H, W)).astype(np.float32))
    res = _randn(r, N, Cc, H, W)
    xq = x.half().float()
    ref = mask * (F.conv2d(xq, w_mod.half().float(), None, padding=1) + bias.view(1, -1, 1, 1)) + \
        (1 - mask) * F.conv2d(xq, w_std.half().float(), None, padding=1) + res
    wp = torch.from_numpy(pack.pack_conv(pack.interleave16(w_std.numpy(), w_mod.numpy()), 2 * Cc)).to(DEV)
    m4 = torch.zeros(N, H, W, 4, dtype=torch.float32, device=DEV); m4[..., 0] = mask[:, 0].to(DEV)
    out = torch.zeros(N, 1, H, W, Cc, dtype=torch.float32, device=DEV)
    resd = res.permute(0, 2, 3, 1).contiguous().unsqueeze(1).to(DEV)
    ops.conv(_to_cl(x).to(DEV), wp, 2 * Cc, Cc, (1, 3, 3), bias=bias.to(DEV), pixscale=m4, ps_stride=4, res=resd, out0=out, mode=1,
             cfg=17 if kern == "halo256" else _cfg(kern, 0, 10))
    torch.cuda.synchronize()
    assert ops.rel_err(_from_cl(out)[:, :, 0], ref) < 2e-3


@pytest.mark.parametrize("kern", KERNELS + ["halo256"])
@pytest.mark.parametrize("xshift", [0, 1])
def test_conv_spade(xshift, kern):
    """gamma/beta convs + instance-norm modulation epilogue == SPADE.forward (util.py:295-302) + leaky_relu(0.2)."""
    import hip_ops as ops
    from canonswap_amd import pack
    r = _rng(14 + xshift)
    N, Cc, S = 2, (128 if kern == "halo256" else 64), 32
    Sx = S >> xshift
    actv = _randn(r, N, 128, S, S)
    x = _randn(r, N, Cc, Sx, Sx) * 2 + 0.5
    wg = _randn(r, Cc, 128, 3, 3, scale=0.02); wb = _randn(r, Cc, 128, 3, 3, scale=0.02)
    bg = _randn(r, Cc, scale=0.1); bb = _randn(r, Cc, scale=0.1)
    xq = x.half().float()
    xn = (xq - xq.mean((2, 3), keepdim=True)) / torch.sqrt(xq.var((2, 3), unbiased=False, keepdim=True) + 1e-5)
    if xshift:
        xn = xn.repeat_interleave(2, 2).repeat_interleave(2, 3)
    aq = actv.half().float()
    ref = F.leaky_relu(xn * (1 + F.conv2d(aq, wg.half().float(), bg, padding=1)) + F.conv2d(aq, wb.half().float(), bb, padding=1), 0.2)
    xd = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    stats = ops.chan_stats(xd.reshape(N, Sx * Sx, Cc))
    wp = torch.from_numpy(pack.pack_conv(pack.interleave16(wg.numpy(), wb.numpy()), 2 * Cc)).to(DEV)
    out = torch.zeros(N, 1, S, S, Cc, dtype=torch.float16, device=DEV)
    ops.conv(_to_cl(actv).to(DEV), wp, 2 * Cc, Cc, (1, 3, 3), bias=bg.to(DEV), bias2=bb.to(DEV), res=xd.unsqueeze(1), res_shift=xshift,
             stats=stats, act0="lrelu", slope0=0.2, out0=out, mode=2, cfg=17 if kern == "halo256" else _cfg(kern, 0, 10))
    torch.cuda.synchronize()
    assert ops.rel_err(_from_cl(out)[:, :, 0], ref) < 3e-3



@pytest.mark.parametrize("cfg,Cc,ep_general", [(17, 256, False), (17, 256, True), (10, 128, False)])
@pytest.mark.parametrize("xshift", [0, 1])
def test_conv_spade_gamma_only(xshift, cfg, Cc, ep_general):
    """cs_conv_desc.mode 5: out0 = act0(IN(x) (1 + conv + bias)) - SPADE's modulation (util.py:295-302) without the beta half, which the
    learned shortcut of a SPADEResnetBlock folds into conv_s (engine.hip spade_shortcut).  128 x 256 tiles: the branch-free copy and the
    general epilogue give the same bits; 128 x 128 tiles only have the general one."""
    import hip_ops as ops
    from canonswap_amd import pack
    r = _rng(140 + xshift + cfg)
    N, S = 2, 32
    Sx = S >> xshift
    actv = F.relu(_randn(r, N, 128, S, S))
    x = _randn(r, N, Cc, Sx, Sx) * 2 + 0.5
    wg = _randn(r, Cc, 128, 3, 3, scale=0.02)
    bg = _randn(r, Cc, scale=0.1)
    xq = x.half().float()
    xn = (xq - xq.mean((2, 3), keepdim=True)) / torch.sqrt(xq.var((2, 3), unbiased=False, keepdim=True) + 1e-5)
    if xshift:
        xn = xn.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = xn * (1 + F.conv2d(actv.half().float(), wg.half().float(), bg, padding=1))
    xd = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    stats = ops.chan_stats(xd.reshape(N, Sx * Sx, Cc))
    wp = torch.from_numpy(pack.pack_conv(wg.numpy(), Cc)).to(DEV)
    outs = []
    for g in ([False, True] if not ep_general else [True]):
        out = torch.zeros(N, 1, S, S, Cc, dtype=torch.float16, device=DEV)
        ops.conv(_to_cl(actv).to(DEV), wp, Cc, Cc, (1, 3, 3), bias=bg.to(DEV), res=xd.unsqueeze(1), res_shift=xshift, stats=stats, out0=out, mode=5, cfg=cfg,
                 ep_general=g)
        torch.cuda.synchronize()
        assert ops.rel_err(_from_cl(out)[:, :, 0], ref) < 3e-3
        outs.append(out.cpu())
    if len(outs) == 2:
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("kern", KERNELS)
def test_conv_pixel_shuffle_sigmoid(kern):
    """conv 64->12 + PixelShuffle(2) + sigmoid (spade_generator.py:36-39,56-57)."""
    import hip_ops as ops
    r = _rng(16)
    N, S = 2, 32
    x = _randn(r, N, 64, S, S)
    w = _randn(r, 12, 64, 3, 3, scale=0.05); b = _randn(r, 12, scale=0.1)
    ref = torch.sigmoid(F.pixel_shuffle(F.conv2d(x.half().float(), w.half().float(), b, padding=1), 2))
    wp = ops.packed_weight(w, 16, DEV)
    b16 = torch.zeros(16); b16[:12] = b
    out = torch.zeros(N, 3, 2 * S, 2 * S, dtype=torch.float32, device=DEV)
    ops.conv(_to_cl(x).to(DEV), wp, 16, 16, (1, 3, 3), bias=b16.to(DEV), act0="sigmoid", out0=out, mode=3, cfg=_cfg(kern, 3, 15))
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max() < 2e-3


@pytest.mark.parametrize("name,N,Cin,Cout,D,H,W,k,ck", [
    ("halo_2d_ck64_db", 2, 256, 128, 1, 32, 32, (1, 3, 3), 64),
    ("halo_2d_ck32_db", 1, 160, 64, 1, 16, 32, (1, 3, 3), 32),
    ("halo_2d_ck64_oddchunks", 1, 96, 128, 1, 16, 16, (1, 3, 3), 64),
    ("halo_3d_multichunk", 1, 144, 64, 4, 16, 16, (3, 3, 3), 32),
    ("halo_7x7x7_mask_like", 1, 144, 32, 4, 16, 16, (7, 7, 7), 32),
    ("halo_1x1", 2, 256, 512, 1, 16, 16, (1, 1, 1), 64),
    ("halo_n16", 1, 512, 16, 1, 16, 16, (1, 3, 3), 64),
    ("halo_256wide_ck64", 2, 128, 256, 1, 32, 32, (1, 3, 3), 64),
    ("halo_256wide_ck32", 1, 96, 512, 1, 16, 32, (1, 3, 3), 32),
    ("halo_static3d_tail_like", 1, 160, 192, 4, 16, 16, (3, 3, 3), 32),
    ("halo_maskp_7x7x1_160", 1, 144, 160, 4, 16, 16, (7, 7, 1), 32),
    ("halo_sk_7x7x7", 1, 144, 32, 4, 16, 16, (7, 7, 7), 32),
    ("halo_sk_3x3x3_multichunk", 2, 96, 32, 8, 8, 8, (3, 3, 3), 32),
    ("halo_sk_fewsteps", 1, 32, 32, 1, 16, 16, (1, 1, 1), 32),
])
def test_conv_halo_variants(name, N, Cin, Cout, D, H, W, k, ck):
    import hip_ops as ops
    r = _rng(zlib.crc32(name.encode()) % 1000)
    x = _randn(r, N, Cin, D, H, W)
    w = _randn(r, Cout, Cin, *k, scale=1.0 / np.sqrt(Cin * np.prod(k)))
    b = _randn(r, Cout, scale=0.1)
    ref = _ref_conv(x, w, b, tuple(kk // 2 for kk in k))
    out = torch.zeros(N, D, H, W, Cout, dtype=torch.float32, device=DEV)
    ops.conv(_to_cl(x).to(DEV), ops.packed_weight(w, Cout, DEV), Cout, Cout, k, bias=b.to(DEV), out0=out, cfg=16 if "_sk_" in name else (18 if "maskp" in name else -2), ck=ck)
    torch.cuda.synchronize()
    err = ops.rel_err(_from_cl(out), ref)
    assert err < 2e-3, (name, err)


def test_grid_sample_3d():
    """Trilinear feature warp == F.grid_sample(align_corners=False) (warping_network.py:46-47), incl. out-of-range points."""
    import hip_ops as ops
    r = _rng(17)
    N, Cc, D, H, W = 2, 32, 16, 16, 16
    x = _randn(r, N, Cc, D, H, W)
    grid = torch.from_numpy(r.uniform(-1.25, 1.25, (N, D, H, W, 3)).astype(np.float32))
    ref = F.grid_sample(x, grid, align_corners=False)
    xd = x.permute(0, 3, 4, 2, 1).contiguous().to(DEV)
    o32, o16 = ops.grid_sample(xd, grid.to(DEV))
    torch.cuda.synchronize()
    assert (o32.cpu().permute(0, 4, 3, 1, 2) - ref).abs().max() < 1e-5
    assert (o16.float().cpu().permute(0, 4, 3, 1, 2) - ref).abs().max() < 4e-3


@pytest.mark.parametrize("dtype,Cc,P", [(torch.float16, 512, 4096), (torch.float32, 32, 65536), (torch.float16, 64, 65536)])
def test_chan_stats(dtype, Cc, P):
    import hip_ops as ops
    r = _rng(18)
    x = (_randn(r, 2, P, Cc) + 0.3).to(dtype)
    st = ops.chan_stats(x.to(DEV)).cpu()
    st2 = ops.chan_stats(x.to(DEV)).cpu()
    assert torch.equal(st, st2)                                         # deterministic (no atomics)
    xf = x.double()
    assert torch.allclose(st[..., 0].double(), xf.mean(1), rtol=1e-5, atol=1e-6)
    assert torch.allclose(st[..., 1].double(), 1.0 / torch.sqrt(xf.var(1, unbiased=False) + 1e-5), rtol=1e-4)


@pytest.mark.parametrize("Cc,P", [(512, 4096), (32, 65536), (64, 1000)])
def test_chan_stats_finish_matches_a_numpy_emulation_of_its_order(Cc, P):
    """chan_stats_finish_kernel: block lane bl (0 .. 63) adds the partial blocks bl, bl + 64, ... in ascending order in fp64, then a pairwise tree over
    the 64 lanes, mean / variance in fp64, one rounding to fp32.  Emulated here from the first pass's partials: equal bit for bit - which pins the order
    independently of how the kernel maps lanes to threads (round 5 changed that mapping, not the order)."""
    import hip_ops as ops
    r = _rng(181)
    N = 3
    x = (_randn(r, N, P, Cc) + 0.3).half()
    st, part = ops.chan_stats(x.to(DEV), with_partials=True)
    torch.cuda.synchronize()
    part = part.cpu().numpy().astype(np.float64)                         # [N, nblk, C, 2]
    nblk = part.shape[1]
    lanes = np.zeros((N, 64, Cc, 2))
    for bl in range(64):
        for b in range(bl, nblk, 64):
            lanes[:, bl] = lanes[:, bl] + part[:, b]
    o = 32
    while o > 0:
        lanes[:, :o] = lanes[:, :o] + lanes[:, o:2 * o]
        o >>= 1
    s, ss = lanes[:, 0, :, 0], lanes[:, 0, :, 1]
    mean = s * (1.0 / P)
    var = np.maximum(ss * (1.0 / P) - mean * mean, 0.0)
    want = np.stack([mean.astype(np.float32), (1.0 / np.sqrt(var + float(np.float32(1e-5)))).astype(np.float32)], axis=-1)
    assert np.array_equal(st.cpu().numpy(), want)


@pytest.mark.parametrize("xmap", [1, 2])
@pytest.mark.parametrize("shape", ["2d_multi_cblk", "3d_v32", "2d_odd_tiles"])
def test_conv_xcd_map_bit_equal(xmap, shape):
    """The XCD-aware workgroup -> (tile, channel block) mappings (ConvParams::xcd_map) only permute which workgroup computes
    which tile: outputs, including the per-tile statistics-free epilogue paths, are bit-identical to the identity mapping."""
    import hip_ops as ops
    r = _rng(77)
    if shape == "2d_multi_cblk":
        N, Cin, Cout, D, H, W, k, cfg, tile = 3, 64, 512, 1, 32, 32, (1, 3, 3), 10, (0, 0)
    elif shape == "3d_v32":
        N, Cin, Cout, D, H, W, k, cfg, tile = 2, 32, 32, 16, 16, 16, (3, 3, 3), 12, (4, 4)
    else:   # 3 x (24 x 8 / 128) tiles: not a multiple of 8
        N, Cin, Cout, D, H, W, k, cfg, tile = 3, 64, 256, 1, 8, 16, (1, 3, 3), 10, (0, 0)
    x = _to_cl(_randn(r, N, Cin, D, H, W)).to(DEV)
    w = _randn(r, Cout, Cin, *k, scale=1.0 / np.sqrt(Cin * np.prod(k)))
    wp = ops.packed_weight(w, Cout, DEV)
    b = _randn(r, Cout, scale=0.1).to(DEV)
    outs = []
    for m in (0, xmap):
        out = torch.zeros(N, D, H, W, Cout, dtype=torch.float32, device=DEV)
        ops.conv(x, wp, Cout, Cout, k, bias=b, act0="relu", out0=out, cfg=cfg, tile=tile, xcd_map=m)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert float(outs[0].abs().max()) > 0
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("k,tile", [((3, 3, 3), (8, 8)), ((7, 7, 1), (2, 8))])
@pytest.mark.parametrize("cfg", [18, 19])
def test_conv_160_wide_tiles(k, tile, cfg):
    """The hourglass tail (3x3x3, 144 -> 144 of 160 packed) and the kw-split mask conv (7x7x1, 144 -> 160) on the 128- and the
    256-position 160-channel tiles; ragged channel count (142 real inputs in 144) and a batch the 256-position tile does not divide."""
    import hip_ops as ops
    r = _rng(31 + cfg)
    N, Cin, Cout, D, H, W = 3, 142, 150, 16, 16, 16
    x = _randn(r, N, Cin, D, H, W)
    w = _randn(r, Cout, Cin, *k, scale=1.0 / np.sqrt(Cin * np.prod(k)))
    b = _randn(r, Cout, scale=0.1)
    ref = F.relu(_ref_conv(x, w, b, tuple(kk // 2 for kk in k)))
    buf = torch.zeros(N, D, H, W, 144, dtype=torch.float16, device=DEV)
    buf[..., :Cin] = _to_cl(x).to(DEV)
    wp = ops.packed_weight(w, 160, DEV)
    b160 = torch.zeros(160); b160[:Cout] = b
    out = torch.zeros(N, D, H, W, 160, dtype=torch.float32, device=DEV)
    ops.conv(buf, wp, 160, 160, k, cin=144, bias=b160.to(DEV), act0="relu", out0=out, cfg=cfg, tile=tile)
    torch.cuda.synchronize()
    assert ops.rel_err(_from_cl(out[..., :Cout]), ref) < 2e-3
    assert float(out[..., Cout:].abs().max()) == 0.0


def test_conv_256x64_tile():
    """3x3x3, 112 -> 64 (first hourglass encoder block) on the 256-position x 64-channel tile against the 128x64 one."""
    import hip_ops as ops
    r = _rng(41)
    N, Cin, Cout, D, H, W = 2, 110, 64, 16, 16, 16
    x = _randn(r, N, Cin, D, H, W)
    w = _randn(r, Cout, Cin, 3, 3, 3, scale=1.0 / np.sqrt(Cin * 27))
    b = _randn(r, Cout, scale=0.1)
    ref = F.relu(_ref_conv(x, w, b, (1, 1, 1)))
    buf = torch.zeros(N, D, H, W, 144, dtype=torch.float16, device=DEV)
    buf[..., 32:32 + Cin] = _to_cl(x).to(DEV)
    wp = ops.packed_weight(w, 64, DEV)
    outs = []
    for cfg in (11, 20):
        out = torch.zeros(N, D, H, W, 64, dtype=torch.float16, device=DEV)
        ops.conv(buf[..., 32:], wp, 64, 64, (3, 3, 3), cin=112, bias=b.to(DEV), act0="relu", out0=out, cfg=cfg, tile=(8, 8))
        torch.cuda.synchronize()
        assert ops.rel_err(_from_cl(out), ref) < 3e-3
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])        # same K order per output element: bit-identical across tile shapes


@pytest.mark.parametrize("N,H,W", [(2, 64, 64), (3, 16, 32), (1, 4, 16)])
def test_t_mask_valu_kernel(N, H, W):
    """T's mask conv (512 -> 1, 3x3, sigmoid) on its VALU kernel against torch and against the same layer on the MFMA conv kernel
    (another summation order: 1e-5 on the gate); only element 0 of each group of four is written."""
    import hip_ops as ops
    r = _rng(77 + H)
    x = F.relu(_randn(r, N, 512, H, W))
    w = _randn(r, 1, 512, 3, 3, scale=2.0 / np.sqrt(512 * 9))
    b = _randn(r, 4, scale=0.1)
    ref = torch.sigmoid(F.conv2d(x.half().float(), w.half().float(), b[:1], padding=1))[:, 0]
    xd = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    wp = ops.packed_weight(w.unsqueeze(2), 16, DEV)
    out = ops.t_mask(xd, wp, b.to(DEV))
    torch.cuda.synchronize()
    assert float((out[..., 0].cpu() - ref).abs().max()) < 2e-5
    assert float((out[..., 1:] + 1.0).abs().max()) == 0.0
    halo = torch.zeros(N, 1, H, W, 4, dtype=torch.float32, device=DEV)
    ops.conv(xd.unsqueeze(1), wp, 16, 4, (1, 3, 3), bias=b.to(DEV), act0="sigmoid", out0=halo, cfg=14)
    torch.cuda.synchronize()
    assert float((out[..., 0] - halo[:, 0, ..., 0]).abs().max()) < 2e-5


def test_t_mask_segment_length_does_not_change_bits():
    """Launches of fewer than 1024 waves (one or two 64 x 64 frames) run t_mask_kernel with 4-column segments, larger ones with 16-column
    segments (kernels.hip, launch_t_mask): a frame alone and the same frame inside a batch of eight give the same bits."""
    import hip_ops as ops
    r = _rng(771)
    x = F.relu(_randn(r, 8, 512, 64, 64))
    w = _randn(r, 1, 512, 3, 3, scale=2.0 / np.sqrt(512 * 9))
    b = _randn(r, 4, scale=0.1)
    xd = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    wp = ops.packed_weight(w.unsqueeze(2), 16, DEV)
    big = ops.t_mask(xd, wp, b.to(DEV))
    for n in (0, 5):
        one = ops.t_mask(xd[n:n + 1].contiguous(), wp, b.to(DEV))
        torch.cuda.synchronize()
        assert torch.equal(one[0], big[n])


@pytest.mark.parametrize("stat", [False, True])
def test_conv_256x64_tile_2d(stat):
    """3x3, 128 -> 64 (G's last up block / F's first down block) on the 2-D 256-position x 64-channel tile (16x16) against the 128x64
    one: the same bits in the output, the same per-(16 x 4 positions) partial statistics (in another block order)."""
    import hip_ops as ops
    r = _rng(43)
    N, Cin, Cout, S = 2, 128, 64, 32
    x = _randn(r, N, Cin, S, S)
    w = _randn(r, Cout, Cin, 3, 3, scale=1.0 / np.sqrt(Cin * 9))
    b = _randn(r, Cout, scale=0.1)
    res = _randn(r, N, Cout, S, S)
    ref = _ref_conv(x.unsqueeze(2), w.unsqueeze(2), b, (0, 1, 1)) + res.half().float().unsqueeze(2)
    wp = ops.packed_weight(w.unsqueeze(2), 64, DEV)
    xd, rd = _to_cl(x).to(DEV), _to_cl(res).to(DEV)
    outs, sums = [], []
    for cfg in (11, 20):
        out = torch.zeros(N, 1, S, S, 64, dtype=torch.float16, device=DEV)
        nblk = S * S // 64
        so = torch.zeros(N, nblk, 64, 2, dtype=torch.float32, device=DEV) if stat else None
        ops.conv(xd, wp, 64, 64, (1, 3, 3), bias=b.to(DEV), res=rd, out0=out, cfg=cfg, tile=(16, 16) if cfg == 20 else (0, 0), stat_out=so)
        torch.cuda.synchronize()
        assert ops.rel_err(_from_cl(out), ref) < 3e-3
        outs.append(out.cpu())
        if stat:
            sums.append(so.double().sum(1).cpu())
    assert torch.equal(outs[0], outs[1])
    if stat:
        o = outs[0].double().reshape(N, S * S, 64)
        assert torch.allclose(sums[0][..., 0], o.sum(1), rtol=1e-5, atol=1e-3) and torch.allclose(sums[1][..., 0], o.sum(1), rtol=1e-5, atol=1e-3)
        assert torch.allclose(sums[0], sums[1], rtol=1e-6, atol=1e-4)


@pytest.mark.parametrize("k,tile,cfg,cout_pad,cin_real,cin", [((3, 3, 3), (8, 8), 19, 160, 142, 144), ((7, 7, 1), (2, 8), 19, 160, 142, 144),
                                                              ((3, 3, 3), (8, 8), 20, 64, 110, 112)])
def test_conv_ragged_last_chunk_paired_taps(k, tile, cfg, cout_pad, cin_real, cin):
    """Cin % 32 == 16 (hourglass tail / mask conv 144, first encoder block 112): the 16 real channels of the last chunk of two taps
    that are neighbours along the row share one 32-deep MFMA step (ConvParams::ragged, weights re-packed by cs_op_pair_ragged as the
    engine does at load time).  Same convolution - against the fp32 reference and against the unpaired launch."""
    import hip_ops as ops
    r = _rng(51 + cfg + k[0])
    N, D, H, W = 3, 16, 16, 16
    Cout = cout_pad - 10 if cout_pad == 160 else cout_pad
    x = _randn(r, N, cin_real, D, H, W)
    w = _randn(r, Cout, cin_real, *k, scale=1.0 / np.sqrt(cin_real * np.prod(k)))
    b = _randn(r, Cout, scale=0.1)
    ref = F.relu(_ref_conv(x, w, b, tuple(kk // 2 for kk in k)))
    buf = torch.zeros(N, D, H, W, cin, dtype=torch.float16, device=DEV)
    buf[..., :cin_real] = _to_cl(x).to(DEV)
    bp = torch.zeros(cout_pad); bp[:Cout] = b
    outs = []
    for ragged in (False, True):
        wp = ops.packed_weight(w, cout_pad, DEV)
        if ragged:
            ops.pair_ragged(wp, cout_pad, cin, k)
        out = torch.zeros(N, D, H, W, cout_pad, dtype=torch.float32, device=DEV)
        ops.conv(buf, wp, cout_pad, cout_pad, k, cin=cin, bias=bp.to(DEV), act0="relu", out0=out, cfg=cfg, tile=tile, ragged=ragged)
        torch.cuda.synchronize()
        assert ops.rel_err(_from_cl(out[..., :Cout]), ref) < 2e-3
        assert Cout == cout_pad or float(out[..., Cout:].abs().max()) == 0.0
        outs.append(out.cpu())
    assert ops.rel_err(outs[1], outs[0]) < 1e-5          # fp32 accumulation, another order within the last chunk


@pytest.mark.parametrize("k,tile,shape,ragged", [((3, 3, 3), (8, 8), (16, 32, 32), True), ((3, 3, 3), (8, 8), (8, 16, 16), False),
                                                 ((7, 7, 1), (4, 8), (8, 16, 32), True)])
def test_conv_256x160_tiles_walk_a_tile_list(k, tile, shape, ragged):
    """The 256 x 160 kernels run persistent (ConvParams::persist_total): a workgroup's first tile is staged by a burst, every later one under
    the last chunk of its predecessor, with the weight ring carried across the epilogue.  More tiles than workgroups, a count that does not
    divide by 8 XCDs: against the fp32 reference, and the same sample at two places of the walk gives the same bits."""
    import hip_ops as ops
    r = _rng(91 + k[0])
    D, H, W = shape
    per = D * H * W // 256
    N = (2 * 256 + 24) // per + 1                      # > 2 tiles per workgroup on 256 CUs, ragged shares
    N += N % 2
    cin_real, cin, cout_pad, Cout = (142, 144, 160, 150) if ragged else (128, 128, 160, 160)
    xh = _randn(r, N // 2, cin_real, D, H, W)
    x = torch.cat([xh, xh], 0)
    w = _randn(r, Cout, cin_real, *k, scale=1.0 / np.sqrt(cin_real * np.prod(k)))
    b = _randn(r, Cout, scale=0.1)
    buf = torch.zeros(N, D, H, W, cin, dtype=torch.float16, device=DEV)
    buf[..., :cin_real] = _to_cl(x).to(DEV)
    bp = torch.zeros(cout_pad); bp[:Cout] = b
    wp = ops.packed_weight(w, cout_pad, DEV)
    if ragged:
        ops.pair_ragged(wp, cout_pad, cin, k)
    out = torch.full((N, D, H, W, cout_pad), float("nan"), dtype=torch.float32, device=DEV)
    ops.conv(buf, wp, cout_pad, cout_pad, k, cin=cin, bias=bp.to(DEV), act0="relu", out0=out, cfg=19, tile=tile, ragged=ragged)
    torch.cuda.synchronize()
    o = out.cpu()
    assert torch.equal(o[: N // 2], o[N // 2:])
    for i in (0, N // 2 - 1):                          # a first tile of a workgroup, a later one
        ref = F.relu(_ref_conv(xh[i:i + 1], w, b, tuple(kk // 2 for kk in k)))
        assert ops.rel_err(_from_cl(o[i:i + 1, ..., :Cout]), ref) < 2e-3
    assert Cout == cout_pad or float(o[..., Cout:].abs().max()) == 0.0


@pytest.mark.parametrize("cfg,N,Cin,Cout,H,W", [(10, 2, 96, 128, 32, 48), (17, 3, 32, 256, 16, 32)])
def test_conv2d_avgpool_in_the_epilogue(cfg, N, Cin, Cout, H, W):
    """DownBlock2d (util.py:150-166): conv3x3 + folded BN + ReLU + AvgPool2d(2) as ONE launch on the 16 x 8 tiles with 32-channel chunks (F's
    down blocks): the h + 1 neighbour of a window is the same lane of the next position block, the w + 1 neighbour a DPP row shift."""
    import hip_ops as ops
    r = _rng(500 + cfg)
    x = F.relu(_randn(r, N, Cin, 1, H, W))
    w = _randn(r, Cout, Cin, 1, 3, 3, scale=1.0 / np.sqrt(9 * Cin))
    b = _randn(r, Cout, scale=0.1)
    ref = F.avg_pool3d(F.relu(_ref_conv(x, w, b, (0, 1, 1))), (1, 2, 2))
    xd = _to_cl(x).to(DEV)
    wp = ops.packed_weight(w, Cout, DEV)
    obuf = torch.full((N, 1, H // 2, W // 2, Cout + 32), -5.0, dtype=torch.float16, device=DEV)
    ops.conv(xd, wp, Cout, Cout, (1, 3, 3), bias=b.to(DEV), act0="relu", out0=obuf[..., 16:16 + Cout], out_dims=(N, 1, H, W), cfg=cfg, pool_hw=True)
    torch.cuda.synchronize()
    assert ops.rel_err(_from_cl(obuf[..., 16:16 + Cout]), ref) < 2e-3
    assert bool((obuf[..., :16] == -5.0).all()) and bool((obuf[..., 16 + Cout:] == -5.0).all())
    full = torch.zeros(N, 1, H, W, Cout, dtype=torch.float32, device=DEV)       # the same conv unpooled, fp32: its 2x2 means are the pooled output
    ops.conv(xd, wp, Cout, Cout, (1, 3, 3), bias=b.to(DEV), act0="relu", out0=full, cfg=cfg)
    torch.cuda.synchronize()
    mean4 = full.view(N, 1, H // 2, 2, W // 2, 2, Cout).mean(dim=(3, 5))
    assert float((obuf[..., 16:16 + Cout].float() - mean4).abs().max()) <= float(mean4.abs().max()) * 2.0 ** -10


def test_conv_split_weights_use_fp16_subnormals():
    """pack._hi_lo: for |w| < 2^-3 the W_lo half of a split-precision weight is an fp16 subnormal.  The MFMA must consume it exactly: the conv
    over [x | x] with [W_hi | W_lo] then matches a float64 conv with the unrounded weights to ~2^-18 relative; flushed to zero it would be
    W_hi alone, 2^-11."""
    import hip_ops as ops
    from canonswap_amd import pack
    r = _rng(77)
    N, Cin, Cout, S = 1, 64, 64, 32
    x = torch.from_numpy(r.standard_normal((N, Cin, S, S)).astype(np.float16))
    w = (0.01 * r.standard_normal((Cout, Cin, 3, 3))).astype(np.float64)                     # |w| ~ 1e-2: every W_lo is subnormal
    hl = pack._hi_lo(w)
    lo = hl[:, Cin:]
    assert np.abs(lo[lo != 0]).max() < 2.0 ** -14                                             # subnormal range of fp16
    ref = F.conv2d(x.double(), torch.from_numpy(w), None, padding=1)
    ref_hi = F.conv2d(x.double(), torch.from_numpy(hl[:, :Cin]), None, padding=1)
    xd = torch.cat([x, x], dim=1).permute(0, 2, 3, 1).contiguous().unsqueeze(1).to(DEV)      # [N, 1, H, W, 2 Cin]
    wp = torch.from_numpy(pack.pack_conv(hl, Cout)).to(DEV)
    out = torch.zeros(N, 1, S, S, Cout, dtype=torch.float32, device=DEV)
    ops.conv(xd, wp, Cout, Cout, (1, 3, 3), out0=out, cfg=-2)
    torch.cuda.synchronize()
    got = out.cpu()[:, 0].permute(0, 3, 1, 2).double()
    err = float((got - ref).norm() / ref.norm()); err_hi = float((ref_hi - ref).norm() / ref.norm())
    assert err_hi > 1e-4                     # what a flush of the subnormals would leave
    assert err < 2e-6, (err, err_hi)         # fp32 accumulation of 1152 products + 2^-18-relative weights


@pytest.mark.parametrize("cfg,tile,N,Cin,Cout,H", [(10, (0, 0), 2, 64, 128, 16), (10, (4, 4), 2, 128, 128, 4), (11, (0, 0), 3, 64, 64, 8), (20, (8, 8), 2, 112, 64, 16),
                                                    (10, (0, 0), 1, 96, 256, 32)])
def test_conv_avgpool_in_the_epilogue(cfg, tile, N, Cin, Cout, H):
    """DownBlock3d (util.py:185-190): conv3x3x3 + folded BN + ReLU + AvgPool3d((1,2,2)) as ONE launch (ConvParams::pool_hw): the average of
    the four fp32 values goes straight into the next level's concat buffer, at a channel offset; neighbouring channels stay untouched."""
    import hip_ops as ops
    r = _rng(400 + cfg + H)
    D, W = 16, H
    x = _randn(r, N, Cin, D, H, W)
    w = _randn(r, Cout, Cin, 3, 3, 3, scale=1.0 / np.sqrt(27 * Cin))
    b = _randn(r, Cout, scale=0.1)
    ref = F.avg_pool3d(F.relu(_ref_conv(x, w, b, 1)), (1, 2, 2))
    xd = _to_cl(x).to(DEV)
    wp = ops.packed_weight(w, Cout, DEV)
    obuf = torch.full((N, D, H // 2, W // 2, Cout + 32), -5.0, dtype=torch.float16, device=DEV)
    ops.conv(xd, wp, Cout, Cout, (3, 3, 3), bias=b.to(DEV), act0="relu", out0=obuf[..., 16:16 + Cout], out_dims=(N, D, H, W), cfg=cfg, tile=tile, pool_hw=True)
    torch.cuda.synchronize()
    assert ops.rel_err(_from_cl(obuf[..., 16:16 + Cout]), ref) < 2e-3
    assert bool((obuf[..., :16] == -5.0).all()) and bool((obuf[..., 16 + Cout:] == -5.0).all())
