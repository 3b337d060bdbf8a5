"""The branch-free copies of the conv epilogue (conv_epilogue.h, CONV_EPILOGUE: one per tensor combination of the engine's hot layers)
against the general epilogue of the same kernel (cs_conv_desc::ep_general): the same bits in every output, for every combination and
tile family that carries a copy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _t(r, *shape, scale=1.0, dtype=torch.float16):
    return torch.from_numpy((scale * r.standard_normal(shape)).astype(np.float32)).to(dtype).to(DEV)


def _wgt(r, cin, cout_pad, k):
    return _t(r, ((cin + 31) // 32) * int(np.prod(k)), cout_pad, 32, scale=0.02)


def _both(run):
    """run(ep_general) -> list of output tensors; compares fast against general bit for bit"""
    a = run(False)
    torch.cuda.synchronize()
    a = [t.clone() for t in a]
    b = run(True)
    torch.cuda.synchronize()
    for i, (x, y) in enumerate(zip(a, b)):
        assert float(x.float().abs().max()) > 0.0, "output %d is empty" % i
        assert torch.equal(x, y), "output %d differs (max abs diff %g)" % (i, float((x.float() - y.float()).abs().max()))


@pytest.mark.parametrize("cfg", [10, 17])            # 128x128 (three workgroups per CU) and 128x256 tiles
@pytest.mark.parametrize("res", ["none", "fp16"])
@pytest.mark.parametrize("stat", [False, True])
def test_std_fp16_out(cfg, res, stat):
    """G's 3x3 convs: fp16 output, optional fp16 residual, with and without the statistics of the stored output"""
    import hip_ops as ops
    r = _rng(100 + cfg + (res == "fp16") + 2 * stat)
    N, S, C = 3, 32, 512
    x = torch.relu(_t(r, N, 1, S, S, C))
    w = _wgt(r, C, C, (1, 3, 3))
    b = _t(r, C, dtype=torch.float32)
    rs = _t(r, N, 1, S, S, C) if res == "fp16" else None
    nblk = (S * S // 128) * 2

    def run(gen):
        out = torch.zeros(N, 1, S, S, C, dtype=torch.float16, device=DEV)
        so = torch.zeros(N, nblk, C, 2, dtype=torch.float32, device=DEV) if stat else None
        ops.conv(x, w, C, C, (1, 3, 3), bias=b, res=rs, out0=out, cfg=cfg, stat_out=so, ep_general=gen)
        return [out] + ([so] if stat else [])
    _both(run)


def test_std_fp32_res_two_outputs():
    """R's 2-D blocks: fp32 residual stream in and out plus the fp16 copy through the next block's affine + LeakyReLU"""
    import hip_ops as ops
    r = _rng(7)
    N, S, C = 2, 32, 512
    x = torch.relu(_t(r, N, 1, S, S, C))
    w = _wgt(r, C, C, (1, 3, 3))
    b = _t(r, C, dtype=torch.float32)
    rs = _t(r, N, 1, S, S, C, dtype=torch.float32)
    s2 = _t(r, C, dtype=torch.float32); t2 = _t(r, C, dtype=torch.float32)

    def run(gen):
        o0 = torch.zeros(N, 1, S, S, C, dtype=torch.float32, device=DEV)
        o1 = torch.zeros(N, 1, S, S, C, dtype=torch.float16, device=DEV)
        ops.conv(x, w, C, C, (1, 3, 3), bias=b, res=rs, out0=o0, out1=o1, s2=s2, t2=t2, act1="lrelu", slope1=0.01, cfg=17, ep_general=gen)
        return [o0, o1]
    _both(run)


def test_std_res_second_output_only():
    """G.up_1.conv_1 (64 -> 64 at 256 x 256 on the 2-D 256x64 tile): fp16 residual, no out0, out1 = leaky_relu(conv + bias + res, 0.2) -
    the copy against the general epilogue, and both against fp32 torch on the fp16-rounded operands"""
    import hip_ops as ops
    from canonswap_amd import pack
    r = _rng(11)
    N, S, C = 2, 64, 64
    x = torch.relu(_t(r, N, 1, S, S, C))
    w32 = (0.05 * r.standard_normal((C, C, 1, 3, 3))).astype(np.float32)
    w = torch.from_numpy(pack.pack_conv(w32, C)).to(DEV)
    b = _t(r, C, dtype=torch.float32)
    rs = _t(r, N, 1, S, S, C)
    got = []

    def run(gen):
        o1 = torch.zeros(N, 1, S, S, C, dtype=torch.float16, device=DEV)
        ops.conv(x, w, C, C, (1, 3, 3), bias=b, res=rs, out1=o1, act1="lrelu", slope1=0.2, cfg=20, tile=(16, 16), ep_general=gen)
        got.append(o1)
        return [o1]
    _both(run)
    wh = torch.from_numpy(w32).half().float()[:, :, 0]
    ref = torch.nn.functional.conv2d(x[:, 0].float().cpu().permute(0, 3, 1, 2), wh, b.cpu(), padding=1) + rs[:, 0].float().cpu().permute(0, 3, 1, 2)
    ref = torch.nn.functional.leaky_relu(ref, 0.2).permute(0, 2, 3, 1)
    err = float((got[0][:, 0].float().cpu() - ref).abs().max())
    assert err < 4e-3 * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("cfg,ck", [(10, 64), (10, 32), (11, 64), (11, 32), (13, 64)])      # 128x128, 128x64, 128x32 tiles of the 1x1 kernels
@pytest.mark.parametrize("form", ["gelu_f32", "f32", "res_f32_inplace", "sigmoid_f32"])
def test_motion_linear_layers(cfg, ck, form):
    """The motion extractor's 1x1 convs (convnextv2.py:39-45): fp32 output behind GELU, fp32 output, fp32 residual added in place - the
    branch-free copies against the general epilogue (sigmoid: no copy exists, both runs take the general path), and against torch"""
    import hip_ops as ops
    from canonswap_amd import pack
    r = _rng(300 + cfg + ck + len(form))
    N, S, Cin = 4, 16, 192
    Cout = {10: 384, 11: 192, 13: 96}[cfg]
    x = _t(r, N, 1, S, S, Cin)
    w32 = (0.08 * r.standard_normal((Cout, Cin, 1, 1, 1))).astype(np.float32)
    w = torch.from_numpy(pack.pack_conv(w32, Cout)).to(DEV)
    b = _t(r, Cout, dtype=torch.float32)
    rs0 = _t(r, N, 1, S, S, Cout, dtype=torch.float32)
    act = {"gelu_f32": "gelu", "sigmoid_f32": "sigmoid"}.get(form, "none")
    got = []

    def run(gen):
        o0 = rs0.clone() if form == "res_f32_inplace" else torch.zeros(N, 1, S, S, Cout, dtype=torch.float32, device=DEV)
        ops.conv(x, w, Cout, Cout, (1, 1, 1), bias=b, act0=act, res=o0 if form == "res_f32_inplace" else None, out0=o0, cfg=cfg, ck=ck, ep_general=gen)
        got.append(o0)
        return [o0]
    _both(run)
    ref = x.float().cpu().reshape(-1, Cin) @ torch.from_numpy(w32).half().float().reshape(Cout, Cin).T + b.cpu()
    ref = {"gelu": torch.nn.functional.gelu, "sigmoid": torch.sigmoid, "none": lambda t: t}[act](ref)
    if form == "res_f32_inplace":
        ref = ref + rs0.cpu().reshape(-1, Cout)
    err = float((got[0].cpu().reshape(-1, Cout) - ref).abs().max())
    assert err < 2e-3 * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("xshift", [0, 1])
def test_spade(xshift):
    """gamma / beta convs with the modulation epilogue on the 128x256 tile (the tensor being modulated at the same or half resolution)"""
    import hip_ops as ops
    r = _rng(20 + xshift)
    N, S, C = 2, 32, 256
    Sx = S >> xshift
    a = torch.relu(_t(r, N, 1, S, S, 128))
    w = _wgt(r, 128, 2 * C, (1, 3, 3))
    bg = _t(r, C, dtype=torch.float32); bb = _t(r, C, dtype=torch.float32)
    xm = _t(r, N, 1, Sx, Sx, C)
    stats = torch.stack([_t(r, N, C, dtype=torch.float32), torch.rand(N, C, device=DEV) + 0.5], dim=2).contiguous()

    def run(gen):
        out = torch.zeros(N, 1, S, S, C, dtype=torch.float16, device=DEV)
        ops.conv(a, w, 2 * C, C, (1, 3, 3), bias=bg, bias2=bb, res=xm, res_shift=xshift, stats=stats, act0="lrelu", slope0=0.2, out0=out,
                 mode=2, cfg=17, ep_general=gen)
        return [out]
    _both(run)


@pytest.mark.parametrize("k,tile,cout", [((3, 3, 3), (8, 8), 144), ((3, 3, 3), (8, 8), 160)])
def test_160_wide_tile(k, tile, cout):
    """hourglass tail on the 256 x 160 tile: 144 of 160 packed channels (the second channel wave takes the ragged copy) and all 160"""
    import hip_ops as ops
    r = _rng(40 + cout)
    N, D, H, W = 2, 16, 16, 16
    x = torch.relu(_t(r, N, D, H, W, 144))
    w = _wgt(r, 144, 160, k)
    b = _t(r, 160, dtype=torch.float32)

    def run(gen):
        out = torch.zeros(N, D, H, W, cout, dtype=torch.float16, device=DEV)
        ops.conv(x, w, 160, cout, k, bias=b, act0="relu", out0=out, cfg=19, tile=tile, ep_general=gen)
        return [out]
    _both(run)


def test_volume_128x128_and_256x64():
    """hourglass encoder / decoder convs (3x3x3 on 8x8x2 tiles) and the first encoder block's 256 x 64 tile"""
    import hip_ops as ops
    r = _rng(9)
    N, D, H, W = 2, 16, 16, 16
    x = torch.relu(_t(r, N, D, H, W, 128))
    for cfg, cin, cout in [(10, 128, 128), (20, 96, 64)]:
        w = _wgt(r, cin, cout, (3, 3, 3))
        b = _t(r, cout, dtype=torch.float32)

        def run(gen):
            out = torch.zeros(N, D, H, W, cout, dtype=torch.float16, device=DEV)
            ops.conv(x[..., :cin], w, cout, cout, (3, 3, 3), cin=cin, bias=b, act0="relu", out0=out, cfg=cfg, tile=(8, 8), ep_general=gen)
            return [out]
        _both(run)
