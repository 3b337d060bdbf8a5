"""vol32.hip (the 3x3x3 32 -> 32 convolutions of the feature volume: ResBlock3d util.py:80-102, ResBlock3D_stage3_leak util.py:528-544)
against (a) plain PyTorch fp32 CPU conv3d of the fp16-rounded operands and (b) the generic LDS-halo kernel it replaces.

Both kernels accumulate the 27 taps of an output element in the same (kd, kh, kw) order, one 32-channel MFMA step each, and apply the
same epilogue formulas in the same order, so (b) is bit-exact: torch.equal.  Shapes cover a whole 64 x 64 volume with an odd batch,
segments cut along H (a batch too small to fill the chip with whole strips), rows fewer than the ring, and W = 8 (one strip, both
halo columns outside)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
CFG_VOL32, CFG_H_256x32 = 30, 12


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _randn(r, *shape, scale=1.0):
    return torch.from_numpy((scale * r.standard_normal(shape)).astype(np.float32))


def _hwdc(t):                # N C D H W -> N H W D C
    return t.permute(0, 3, 4, 2, 1).contiguous()


def _view(t):                # N H W D C storage -> logical [N, D, H, W, C] view
    return t.permute(0, 3, 1, 2, 4)


def _back(t):                # N H W D C -> N C D H W fp32 cpu
    return t.float().cpu().permute(0, 4, 3, 1, 2)


def _run(cfg, x16, wp, b, res, s2, t2, first):
    """first: conv1 of a ResBlock3d (fp16 output, ReLU); else conv2 (fp32 residual stream in / out + pre-activated fp16 copy)."""
    import hip_ops as ops
    N, H, W, D, Cc = x16.shape
    kw = dict(tile=(4, 4)) if cfg == CFG_H_256x32 else {}
    if first:
        out = torch.full((N, H, W, D, Cc), 3.0, dtype=torch.float16, device=DEV)
        ops.conv(_view(x16), wp, 32, 32, (3, 3, 3), bias=b, act0="relu", out0=_view(out), cfg=cfg, **kw)
        torch.cuda.synchronize()
        return out, None
    out0 = torch.full((N, H, W, D, Cc), 3.0, dtype=torch.float32, device=DEV)
    out1 = torch.full((N, H, W, D, Cc), 3.0, dtype=torch.float16, device=DEV)
    ops.conv(_view(x16), wp, 32, 32, (3, 3, 3), bias=b, res=_view(res), out0=_view(out0), s2=s2, t2=t2, act1="relu", out1=_view(out1),
             cfg=cfg, **kw)
    torch.cuda.synchronize()
    return out0, out1


@pytest.mark.parametrize("N,H,W", [(3, 64, 64), (1, 64, 64), (2, 24, 16), (5, 2, 8), (1, 1, 8), (2, 9, 24), (32, 64, 64), (2, 7, 8)])
@pytest.mark.parametrize("first", [True, False])
def test_vol32_equals_halo_kernel_and_torch(N, H, W, first):
    import hip_ops as ops
    r = _rng(1000 * N + 10 * H + W + int(first))
    Cc, D = 32, 16
    x = _randn(r, N, Cc, D, H, W)
    x = torch.relu(x) if first else x                         # conv1 sees post-ReLU activations, conv2 any sign
    res = _randn(r, N, Cc, D, H, W)
    w = _randn(r, Cc, Cc, 3, 3, 3, scale=0.04)
    b = _randn(r, Cc, scale=0.1)
    s2 = torch.from_numpy(r.uniform(0.5, 1.5, Cc).astype(np.float32)); t2 = _randn(r, Cc, scale=0.2)
    x16 = _hwdc(x).half().to(DEV)
    resd = _hwdc(res).to(DEV)
    wp = ops.packed_weight(w, 32, DEV)
    args = (x16, wp, b.to(DEV), resd, s2.to(DEV), t2.to(DEV), first)
    new0, new1 = _run(CFG_VOL32, *args)
    halo_ok = H % min(H, 4) == 0 and W % 4 == 0                # the halo kernel tiles H x W in 4 x 4 (vol32 takes any H)
    old0, old1 = _run(CFG_H_256x32, *args) if halo_ok else (None, None)
    y = F.conv3d(x.half().float(), w.half().float(), b, padding=1)
    if first:
        assert ops.rel_err(_back(new0), F.relu(y)) < 2e-3
    else:
        y = y + res
        assert ops.rel_err(_back(new0), y) < 2e-3
        assert ops.rel_err(_back(new1), F.relu(y * s2.view(1, -1, 1, 1, 1) + t2.view(1, -1, 1, 1, 1))) < 3e-3
        assert not halo_ok or torch.equal(new1, old1)
    assert not halo_ok or torch.equal(new0, old0), float((new0.float() - old0.float()).abs().max())


def test_vol32_strided_views_and_untouched_neighbours():
    """Output into every other sample of a larger buffer and input from a sample-strided view: the kernel honours the N / H / W strides
    (a column of 16 voxels x 32 channels stays contiguous) and writes nothing else."""
    import hip_ops as ops
    r = _rng(77)
    N, H, W, D, Cc = 2, 16, 16, 16, 32
    x = _randn(r, N, Cc, D, H, W)
    w = _randn(r, Cc, Cc, 3, 3, 3, scale=0.04)
    b = _randn(r, Cc, scale=0.1)
    big_in = torch.zeros(2 * N, H, W, D, Cc, dtype=torch.float16, device=DEV)
    big_in[1::2] = _hwdc(x).half().to(DEV)
    big_in[0::2] = 9.0
    big_out = torch.full((2 * N, H, W, D, Cc), -7.0, dtype=torch.float16, device=DEV)
    wp = ops.packed_weight(w, 32, DEV)
    ops.conv(_view(big_in[1::2]), wp, 32, 32, (3, 3, 3), bias=b.to(DEV), act0="relu", out0=_view(big_out[0::2]), cfg=CFG_VOL32)
    torch.cuda.synchronize()
    ref = F.relu(F.conv3d(x.half().float(), w.half().float(), b, padding=1))
    assert ops.rel_err(_back(big_out[0::2]), ref) < 2e-3
    assert bool((big_out[1::2] == -7.0).all())


def test_engine_resblocks_use_vol32_and_match_halo_path(state_dicts, monkeypatch):
    """The appearance feature extractor (12 of these convolutions) through the engine with the kernel on (default) and off
    (CANONSWAP_VOL32=0 is read once per process, so the second engine runs in a subprocess): identical bits."""
    import os
    import subprocess
    import sys
    from canonswap_amd import synth
    from canonswap_amd.can_swap_e2e import can_swapper
    inp = synth.make_frame_inputs(2, seed=31, size=256)
    img = torch.from_numpy(inp["img"])
    sw = can_swapper(None, state_dicts=state_dicts, max_batch=2)
    got = sw.extract_feature_3d(img.cuda()).cpu()
    code = ("import torch, sys; sys.path.insert(0, %r); from canonswap_amd import synth; from canonswap_amd.can_swap_e2e import can_swapper;"
            "sd = synth.to_torch(synth.make_state_dicts(0)); sw = can_swapper(None, state_dicts=sd, max_batch=2);"
            "img = torch.from_numpy(synth.make_frame_inputs(2, seed=31, size=256)['img']);"
            "torch.save(sw.extract_feature_3d(img.cuda()).cpu(), sys.argv[1])") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = "/tmp/vol32_off.pt"
    env = dict(os.environ, CANONSWAP_VOL32="0")
    r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = torch.load(path)
    assert torch.equal(got, want)


@pytest.mark.parametrize("N,H,W", [(2, 64, 64), (1, 16, 8), (3, 24, 16)])
def test_vol32_split_precision_with_statistics(N, H, W):
    """The split-precision form of R's GroupNorm convolutions (util.py:528-544; ConvParams::hilo): activations [hi | lo], weights
    [W_hi | W_lo | W_hi], out = W_hi x_hi + W_lo x_hi + W_hi x_lo accumulated in that order - bit-identical to the halo kernel - and
    good to ~1e-6 against an fp64 convolution of the UNROUNDED operands (what the split buys).  Partial statistics: per (column pair,
    8 rows) sums that must add up to the sums of the stored output."""
    import hip_ops as ops
    r = _rng(5000 + N + H + W)
    Cc, D = 32, 16
    x = _randn(r, N, Cc, D, H, W)
    w = _randn(r, Cc, Cc, 3, 3, 3, scale=0.04)
    b = _randn(r, Cc, scale=0.1)
    xh = _hwdc(x)
    hi = xh.half()
    lo = (xh - hi.float()).half()
    x2 = torch.cat([hi, lo], dim=-1).contiguous().to(DEV)                       # [N][H][W][D][64]
    whi = w.half().float()
    wlo = (w - whi).half().float()
    wp = ops.packed_weight(torch.cat([whi, wlo, whi], dim=1), 32, DEV)            # Cin = 96: three 32-channel chunks
    outs = {}
    for cfg in (CFG_VOL32, CFG_H_256x32):
        out = torch.full((N, H, W, D, Cc), 3.0, dtype=torch.float32, device=DEV)
        kw = dict(tile=(4, 4)) if cfg == CFG_H_256x32 else {}
        nblk = ((H + 7) // 8) * (W // 2)
        part = torch.zeros(N, nblk, Cc, 2, dtype=torch.float32, device=DEV) if cfg == CFG_VOL32 else None
        ops.conv(_view(x2), wp, 32, 32, (3, 3, 3), cin=96, bias=b.to(DEV), out0=_view(out), cfg=cfg, hilo=True, stat_out=part, **kw)
        torch.cuda.synchronize()
        outs[cfg] = (out, part)
    new, part = outs[CFG_VOL32]
    old, _ = outs[CFG_H_256x32]
    assert torch.equal(new, old), float((new - old).abs().max())      # same 81 MFMA steps per element in the same order
    ref = F.conv3d(x.double(), w.double(), b.double(), padding=1)
    assert ops.rel_err(_back(new).double(), ref) < 3e-6
    # statistics: block (h // 8, w // 2) of sample n holds sum / sum of squares over its 2 columns x 8 rows x 16 depth slices
    o = new.double().cpu()                                                       # [N][H][W][D][C]
    Hb = (H + 7) // 8
    want = torch.zeros(N, Hb, W // 2, Cc, 2, dtype=torch.float64)
    for hb in range(Hb):
        blk = o[:, hb * 8:(hb + 1) * 8].reshape(N, -1, W // 2, 2, D, Cc)
        want[:, hb, :, :, 0] = blk.sum(dim=(1, 3, 4))
        want[:, hb, :, :, 1] = (blk ** 2).sum(dim=(1, 3, 4))
    got = part.double().cpu().view(N, Hb, W // 2, Cc, 2)
    assert float((got - want).abs().max() / want.abs().max()) < 1e-5


@pytest.mark.parametrize("kind,with_res", [(1, False), (2, False), (2, True)])
@pytest.mark.parametrize("N,H,W", [(2, 64, 64), (3, 16, 8)])
def test_vol32_transform_staging(kind, with_res, N, H, W):
    """Transform staging (ConvParams::xf_*): the GroupNorm apply + residual + LeakyReLU of util.py:531-540 (kind 2) or the plain hi / lo split
    (kind 1) happens while the split-precision conv stages its input.  Reference: the same transform in torch fp32, split into [hi | lo] and
    run through the DMA-staged kernel; the transformed tensor written back (the new residual stream) against torch.  The transform itself
    is checked bit for bit at engine level (test below); here 1 ulp of the fp32 transform may differ, so 2e-6."""
    import hip_ops as ops
    r = _rng(9000 + 10 * kind + int(with_res) + N + H + W)
    Cc, D = 32, 16
    y = _randn(r, N, H, W, D, Cc)
    res = _randn(r, N, H, W, D, Cc)
    w = _randn(r, Cc, Cc, 3, 3, 3, scale=0.04)
    b = _randn(r, Cc, scale=0.1)
    mean, rstd = _randn(r, N, Cc, scale=0.3), torch.from_numpy(r.uniform(0.5, 2.0, (N, Cc)).astype(np.float32))
    gamma, beta = torch.from_numpy(r.uniform(0.5, 1.5, Cc).astype(np.float32)), _randn(r, Cc, scale=0.2)
    stats = torch.stack([mean, rstd], dim=2).contiguous()
    if kind == 2:
        a = (y - mean.view(N, 1, 1, 1, Cc)) * rstd.view(N, 1, 1, 1, Cc) * gamma + beta + (res if with_res else 0)
        a = F.leaky_relu(a, 0.01)
    else:
        a = y.clone()
    hi = a.half(); lo = (a - hi.float()).half()
    x2 = torch.cat([hi, lo], dim=-1).contiguous().to(DEV)
    whi = w.half().float(); wlo = (w - whi).half().float()
    wp = ops.packed_weight(torch.cat([whi, wlo, whi], dim=1), 32, DEV)
    nblk = ((H + 7) // 8) * (W // 2)

    def run(xf):
        out = torch.full((N, H, W, D, Cc), 3.0, dtype=torch.float32, device=DEV)
        part = torch.zeros(N, nblk, Cc, 2, dtype=torch.float32, device=DEV)
        ops.conv(_view(x2), wp, 32, 32, (3, 3, 3), cin=96, bias=b.to(DEV), out0=_view(out), cfg=CFG_VOL32, hilo=True, stat_out=part, xf=xf)
        torch.cuda.synchronize()
        return out, part

    ref, ref_part = run(None)
    yd, resd = y.to(DEV), res.to(DEV)
    wb = torch.full((N, H, W, D, Cc), -9.0, dtype=torch.float32, device=DEV)
    xf = dict(kind=kind, y=yd)
    if kind == 2:
        xf.update(stats=stats.to(DEV), gamma=gamma.to(DEV), beta=beta.to(DEV), slope=0.01, res=resd if with_res else None, out=wb)
    got, got_part = run(xf)
    assert ops.rel_err(got, ref) < 2e-6
    assert float((got_part - ref_part).abs().max() / ref_part.abs().max()) < 1e-5
    if kind == 2:
        assert float((wb.cpu() - a).abs().max()) < 1e-5         # every voxel of the new residual stream written, once
    assert bool((yd.cpu() == y).all())


def test_engine_refine_with_and_without_transform_staging(state_dicts):
    """R (adaptive_modulate.py:721-733) with the GroupNorm apply inside the conv staging (default) against the stand-alone norm_act / split16
    passes (CANONSWAP_VOL32_XF=0, second engine in a subprocess): both use the same fixed operation sequence (common.h gn_lrelu), the
    convs and their statistics are the same kernels - identical bits."""
    import os
    import subprocess
    import sys
    from canonswap_amd.can_swap_e2e import can_swapper
    r = _rng(21)
    f = torch.from_numpy((0.08 * r.standard_normal((2, 32, 16, 64, 64))).astype(np.float32))
    torch.save(f, "/tmp/xf_in.pt")
    sw = can_swapper(None, state_dicts=state_dicts, max_batch=2)
    got = sw.refine_module(f.cuda()).cpu()
    code = ("import torch, sys; sys.path.insert(0, %r); from canonswap_amd import synth; from canonswap_amd.can_swap_e2e import can_swapper;"
            "sd = synth.to_torch(synth.make_state_dicts(0)); sw = can_swapper(None, state_dicts=sd, max_batch=2);"
            "torch.save(sw.refine_module(torch.load('/tmp/xf_in.pt').cuda()).cpu(), sys.argv[1])") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CANONSWAP_VOL32_XF="0")
    rr = subprocess.run([sys.executable, "-c", code, "/tmp/xf_off.pt"], env=env, capture_output=True, text=True, timeout=600)
    assert rr.returncode == 0, rr.stderr[-2000:]
    want = torch.load("/tmp/xf_off.pt")
    assert torch.equal(got, want), float((got - want).abs().max())


def _resblock_fused(a16, x32, w1p, w2p, b1, b2, s2, t2, act1):
    import ctypes as C
    from canonswap_amd import _lib
    lib = _lib.load()
    N, H, W, D, Cc = a16.shape
    out0 = torch.full((N, H, W, D, Cc), 3.0, dtype=torch.float32, device=DEV)
    out1 = torch.full((N, H, W, D, Cc), 3.0, dtype=torch.float16, device=DEV)
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.cs_op_resblock3d(p(a16), p(x32), p(out0), p(out1), N, H, W, p(w1p), p(w2p), p(b1), p(b2), p(s2), p(t2),
                                    {"none": 0, "relu": 1}[act1], 0.0, st), "cs_op_resblock3d")
    torch.cuda.synchronize()
    return out0, out1


@pytest.mark.parametrize("N,H,W", [(3, 64, 64), (32, 64, 64), (1, 64, 64), (2, 64, 64), (1, 32, 24), (2, 24, 16), (5, 2, 8), (1, 1, 8), (2, 9, 24)])
@pytest.mark.parametrize("post", [True, False])
def test_fused_resblock3d_equals_two_launches(N, H, W, post):
    """vol32_fused.hip: a whole ResBlock3d (util.py:80-102) per launch against conv1 -> conv2 as two vol32 launches (which equal the halo
    kernel bit for bit, tests above): torch.equal on both outputs, and against torch fp32 at the usual tolerance.  One or two frames of a
    64 x 64 volume run 2- / 4-row segments (launch_vol32_fused): the decomposition does not show in the bits."""
    import hip_ops as ops
    r = _rng(3000 + 7 * N + 3 * H + W + int(post))
    Cc, D = 32, 16
    x = _randn(r, N, Cc, D, H, W)
    a = torch.relu(_randn(r, N, Cc, D, H, W))
    w1 = _randn(r, Cc, Cc, 3, 3, 3, scale=0.04); w2 = _randn(r, Cc, Cc, 3, 3, 3, scale=0.04)
    b1 = _randn(r, Cc, scale=0.1); b2 = _randn(r, Cc, scale=0.1)
    s2 = torch.from_numpy(r.uniform(0.5, 1.5, Cc).astype(np.float32)) if post else None
    t2 = _randn(r, Cc, scale=0.2) if post else None
    a16 = _hwdc(a).half().to(DEV); x32 = _hwdc(x).to(DEV)
    w1p, w2p = ops.packed_weight(w1, 32, DEV), ops.packed_weight(w2, 32, DEV)
    dv = lambda t: None if t is None else t.to(DEV)
    got0, got1 = _resblock_fused(a16, x32, w1p, w2p, dv(b1), dv(b2), dv(s2), dv(t2), "relu" if post else "none")
    # two launches
    h, _ = _run(CFG_VOL32, a16, w1p, dv(b1), None, None, None, True)
    out0 = torch.full((N, H, W, D, Cc), 3.0, dtype=torch.float32, device=DEV)
    out1 = torch.full((N, H, W, D, Cc), 3.0, dtype=torch.float16, device=DEV)
    ops.conv(_view(h), w2p, 32, 32, (3, 3, 3), bias=dv(b2), res=_view(x32), out0=_view(out0), s2=dv(s2), t2=dv(t2),
             act1="relu" if post else "none", out1=_view(out1), cfg=CFG_VOL32)
    torch.cuda.synchronize()
    assert torch.equal(got0, out0), float((got0 - out0).abs().max())
    assert torch.equal(got1, out1)
    hh = F.relu(F.conv3d(a.half().float(), w1.half().float(), b1, padding=1)).half().float()
    y = F.conv3d(hh, w2.half().float(), b2, padding=1) + x
    assert ops.rel_err(_back(got0), y) < 2e-3


@pytest.mark.parametrize("B", [3, 1])
def test_engine_feature_extractor_fused_vs_two_launches(state_dicts, B):
    """F (appearance_feature_extractor.py:38-48: six ResBlock3d) and T's six through the engine with the fused kernel (default, at every
    batch size) and with two launches per block (CANONSWAP_VOL32_FUSED=0, subprocess): identical bits."""
    import os
    import subprocess
    import sys
    from canonswap_amd import synth
    from canonswap_amd.can_swap_e2e import can_swapper
    inp = synth.make_frame_inputs(B, seed=33, size=256)
    img = torch.from_numpy(inp["img"])
    idv = torch.from_numpy(synth.make_identity(7))
    sw = can_swapper(None, state_dicts=state_dicts, max_batch=B)
    f = sw.extract_feature_3d(img.cuda())
    got = torch.cat([f.cpu(), sw.swap_module(f, idv.cuda()).cpu()])
    code = ("import torch, sys; sys.path.insert(0, %r); from canonswap_amd import synth; from canonswap_amd.can_swap_e2e import can_swapper;"
            "sd = synth.to_torch(synth.make_state_dicts(0)); sw = can_swapper(None, state_dicts=sd, max_batch=%d);"
            "img = torch.from_numpy(synth.make_frame_inputs(%d, seed=33, size=256)['img']); idv = torch.from_numpy(synth.make_identity(7));"
            "f = sw.extract_feature_3d(img.cuda()); torch.save(torch.cat([f.cpu(), sw.swap_module(f, idv.cuda()).cpu()]), sys.argv[1])"
            ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), B, B)
    env = dict(os.environ, CANONSWAP_VOL32_FUSED="0")
    r = subprocess.run([sys.executable, "-c", code, "/tmp/fused_off.pt"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = torch.load("/tmp/fused_off.pt")
    assert torch.equal(got, want), float((got - want).abs().max())
