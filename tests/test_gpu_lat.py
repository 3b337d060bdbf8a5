"""conv_lat (single-frame kernel of the 512-channel 3x3 layers: 16 x 8 tiles, the K loop split over the three kernel rows across twelve
waves, conv_lat.hip) against conv_halo on the same operands.  The kernel adds the products of an output element in another order (kernel
row -> chunk -> kw -> half, rows summed (g0 + g1) + g2), so the comparison is a tolerance, not torch.equal: the two fp32 sums of 4608
fp16 x fp16 products differ by a few ulp of the fp32 accumulator (relative L2 error < 2e-6 before the fp16 store), and each is compared
with plain PyTorch fp32 on the fp16-rounded operands at the usual operator tolerance.  The forms are the engine's: T blend conv1 / conv2
(adaptive_modulate.py:128-193, 337-349), R's 2-D pair (util.py:120-128), G's 3x3 convs with the next InstanceNorm's statistics
(util.py:329-344).  Deterministic: two runs give the same bits."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_wide import _rng, _tblend_inputs, _run as _run_tblend

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
CFG_HALO_128x128, CFG_HALO_128x64, CFG_LAT = 10, 11, 32


def _std_inputs(seed, N, H, W, Cin, Cout):
    from canonswap_amd import pack
    r = _rng(seed)
    x = np.maximum(r.standard_normal((N, 1, H, W, Cin)), 0).astype(np.float16)
    w = (r.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    d = dict(x=torch.from_numpy(x).to(DEV), w=torch.from_numpy(w), wp=torch.from_numpy(pack.pack_conv(w, Cout)).to(DEV),
             bias=torch.from_numpy((0.1 * r.standard_normal(Cout)).astype(np.float32)).to(DEV),
             res32=torch.from_numpy(r.standard_normal((N, 1, H, W, Cout)).astype(np.float32)).to(DEV),
             s2=torch.from_numpy(r.uniform(0.5, 1.5, Cout).astype(np.float32)).to(DEV),
             t2=torch.from_numpy((0.2 * r.standard_normal(Cout)).astype(np.float32)).to(DEV))
    d["res16"] = d["res32"].half()
    return d


def _rel(a, b):
    a = a.float(); b = b.float()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("form", ["conv1", "conv2"])
@pytest.mark.parametrize("N,H,W", [(1, 64, 64), (2, 32, 32), (1, 8, 16), (3, 24, 48)])
def test_lat_tblend_vs_halo(form, N, H, W):
    d = _tblend_inputs(300 + N + H, N, H, W, 512, 512)
    a = _run_tblend(d, form, CFG_HALO_128x128, 512)
    b = _run_tblend(d, form, CFG_LAT, 512)
    b2 = _run_tblend(d, form, CFG_LAT, 512)
    for u, v, v2 in zip(a, b, b2):
        assert torch.equal(v, v2)                                     # fixed summation order
        assert not bool((v.float() == -7.0).all())
        tol = 2e-6 if v.dtype == torch.float32 else 6e-4             # fp16 stores: one output ulp where the fp32 sums straddle a rounding boundary
        assert _rel(v, u) < tol, (form, N, _rel(v, u))
        assert float((v.float() - u.float()).abs().max()) <= (1e-4 if v.dtype == torch.float32 else 2.0 ** -8 * float(u.float().abs().max()))


def test_lat_tblend_vs_torch():
    """== AdaptiveSharedWeightConv2d's blend (adaptive_modulate.py:139-186) in fp32 on the fp16-rounded operands"""
    import hip_ops as ops
    N, H, W, C = 1, 64, 64, 512
    d = _tblend_inputs(3, N, H, W, C, C)
    out0, out1 = _run_tblend(d, "conv2", CFG_LAT, C)
    xq = d["x"].float().cpu()[:, 0].permute(0, 3, 1, 2)
    q = lambda w: torch.from_numpy(w).half().float()
    m = torch.from_numpy(d["mask"]).unsqueeze(1)
    ref = m * (F.conv2d(xq, q(d["w_mod"]), None, padding=1) + d["bias"].cpu().view(1, -1, 1, 1)) + (1 - m) * F.conv2d(xq, q(d["w_std"]), None, padding=1)
    ref = ref + d["res"].cpu()[:, 0].permute(0, 3, 1, 2)
    got = out0.cpu()[:, 0].permute(0, 3, 1, 2)
    assert ops.rel_err(got, ref) < 1e-3
    ref1 = F.relu(ref * d["s2"].cpu().view(1, -1, 1, 1) + d["t2"].cpu().view(1, -1, 1, 1))
    assert ops.rel_err(out1.float().cpu()[:, 0].permute(0, 3, 1, 2), ref1) < 2e-3


@pytest.mark.parametrize("form", ["plain", "res32_two_outputs", "stat", "stat_res16"])
@pytest.mark.parametrize("N,H,W,Cout", [(1, 64, 64, 512), (2, 32, 32, 512), (1, 16, 32, 256)])
def test_lat_std_forms_vs_halo(form, N, H, W, Cout):
    import hip_ops as ops
    Cin = 512
    d = _std_inputs(431 + N + H, N, H, W, Cin, Cout)
    nblk = (W // 16) * (H // 8) * 2
    outs = []
    for cfg in (CFG_HALO_128x64, CFG_LAT, CFG_LAT):
        o0_16 = torch.full((N, 1, H, W, Cout), -7.0, dtype=torch.float16, device=DEV)
        o0_32 = torch.full((N, 1, H, W, Cout), -7.0, dtype=torch.float32, device=DEV)
        o1 = torch.full((N, 1, H, W, Cout), -7.0, dtype=torch.float16, device=DEV)
        so = torch.full((N * nblk * Cout * 2,), float('nan'), dtype=torch.float32, device=DEV)
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
    for u, v, v2 in zip(*outs):
        assert not bool(torch.isnan(v.float()).any()), form          # every partial-statistics slot was written
        assert torch.equal(v, v2)
        assert _rel(v, u) < (2e-6 if (v.dtype == torch.float32 and v.numel() != N * nblk * Cout * 2) else 6e-4), (form, _rel(v, u))
    # against plain PyTorch fp32 on the fp16-rounded operands
    xq = d["x"].float().cpu()[:, 0].permute(0, 3, 1, 2)
    y = F.conv2d(xq, d["w"].half().float(), d["bias"].cpu(), padding=1)
    got = outs[1][0].float().cpu()[:, 0].permute(0, 3, 1, 2)
    if form == "plain":
        y = F.leaky_relu(y, 0.01)
    elif form == "res32_two_outputs":
        y = y + d["res32"].cpu()[:, 0].permute(0, 3, 1, 2)
    elif form == "stat_res16":
        y = y + d["res16"].float().cpu()[:, 0].permute(0, 3, 1, 2)
    assert ops.rel_err(got, y) < 2e-3, form
    if form.startswith("stat"):      # the partials are sums of the STORED fp16 values over 64 positions, in the 16 x 8 tiles' block order
        so = outs[1][1].cpu().view(N, nblk, Cout, 2).double()
        v = outs[1][0].float().cpu()[:, 0].double()                  # [N, H, W, C]
        want = torch.stack([v.sum(dim=(1, 2)), (v * v).sum(dim=(1, 2))], dim=-1)
        assert torch.allclose(so.sum(dim=1), want, rtol=1e-5, atol=1e-3)


def test_lat_refuses_what_it_does_not_cover():
    d = _tblend_inputs(5, 1, 32, 32, 256, 256)          # Cin = 256: the kernel's eight 64-channel chunks are a compile-time constant
    with pytest.raises(RuntimeError):
        _run_tblend(d, "conv1", CFG_LAT, 256)
