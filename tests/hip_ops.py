"""Helpers that drive the operator-level C ABI (cs_op_*) from torch tensors; used by the -m gpu tests."""
import ctypes as C

import numpy as np
import torch

from canonswap_amd import _lib, pack

ACT = {"none": 0, "relu": 1, "lrelu": 2, "sigmoid": 3, "gelu": 4}


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def strides_cl(t):
    """t: channels-last tensor [N, D, H, W, C] (possibly a strided view); element strides (sN, sD, sH, sW)."""
    assert t.stride(-1) == 1
    return t.stride(0), t.stride(1), t.stride(2), t.stride(3)


def conv(x, wpacked, cout_pad, cout, k, *, cin=None, bias=None, bias2=None, act0="none", slope0=0.0, res=None, res_shift=0,
         pixscale=None, ps_stride=1, out0=None, s2=None, t2=None, act1="none", slope1=0.0, out1=None, stats=None,
         mode=0, cfg=-1, up_shift=0, tile=(0, 0), out_dims=None, ck=0, xcd_map=None, ragged=False, hilo=False, stat_out=None, xf=None, ep_general=False, pool_hw=False):
    """x: [N, D, H, W, C] fp16 view (C contiguous). out0/out1/res: 5-D channels-last views. k = (KD, KH, KW)."""
    lib = _lib.load()
    d = _lib.ConvDesc()
    N, D, H, W = out_dims if out_dims is not None else x.shape[:4]
    d.in_ = x.data_ptr()
    d.in_sN, d.in_sD, d.in_sH, d.in_sW = strides_cl(x)
    d.N, d.D, d.H, d.W = N, D, H, W
    d.Cin = cin if cin is not None else x.shape[4]
    d.up_shift = up_shift
    d.KD, d.KH, d.KW = k
    d.wgt = wpacked.data_ptr()
    d.Cout_pad, d.Cout = cout_pad, cout
    d.bias = 0 if bias is None else bias.data_ptr()
    d.bias2 = 0 if bias2 is None else bias2.data_ptr()
    d.act0, d.slope0 = ACT[act0], slope0
    if res is not None:
        d.res = res.data_ptr(); d.res_f32 = int(res.dtype == torch.float32); d.res_shift = res_shift
        d.res_sN, d.res_sD, d.res_sH, d.res_sW = strides_cl(res)
    d.pixscale = 0 if pixscale is None else pixscale.data_ptr()
    d.ps_stride = ps_stride
    if out0 is not None:
        d.out0 = out0.data_ptr(); d.out0_f32 = int(out0.dtype == torch.float32)
        if mode != 3:
            d.out0_sN, d.out0_sD, d.out0_sH, d.out0_sW = strides_cl(out0)
    d.s2 = 0 if s2 is None else s2.data_ptr()
    d.t2 = 0 if t2 is None else t2.data_ptr()
    d.act1, d.slope1 = ACT[act1], slope1
    if out1 is not None:
        d.out1 = out1.data_ptr()
        d.out1_sN, d.out1_sD, d.out1_sH, d.out1_sW = strides_cl(out1)
    d.stats = 0 if stats is None else stats.data_ptr()
    d.mode, d.cfg = mode, cfg
    d.tile_w, d.tile_h = tile
    d.ck = ck
    d.xcd_map = 0 if xcd_map is None else xcd_map + 1
    d.ragged = int(ragged)
    d.hilo = int(hilo)
    d.ep_general = int(ep_general)
    d.pool_hw = int(pool_hw)
    d.stat_out = 0 if stat_out is None else stat_out.data_ptr()
    if xf is not None:      # dict(kind, y, res, out, stats, gamma, beta, slope): fp32 volumes laid out like out0
        d.xf_kind = xf["kind"]
        for k in ("y", "res", "out", "stats", "gamma", "beta"):
            setattr(d, "xf_" + k, 0 if xf.get(k) is None else xf[k].data_ptr())
        d.xf_slope = xf.get("slope", 0.0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if -1 <= cfg <= 3:        # the independently written cross-check kernel (tests/csrc/conv_igemm.hip, test-only library)
        tl = _lib.load_test_lib()
        if tl.cs_test_conv_igemm(C.byref(d), st) != 0:
            raise RuntimeError("cs_test_conv_igemm failed: " + tl.cs_test_last_error().decode(errors="replace"))
        return
    _lib.check(lib.cs_op_conv(C.byref(d), st), "cs_op_conv")


def t_mask(x, wpacked, bias):
    """x: [N, H, W, 512] fp16 -> sigmoid(conv3x3(x) + bias[0]) as [N, H, W] fp32 (cs_op_t_mask writes at stride 4)"""
    lib = _lib.load()
    N, H, W, _ = x.shape
    out = torch.full((N, H, W, 4), -1.0, dtype=torch.float32, device=x.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.cs_op_t_mask(_p(x), _p(wpacked), _p(bias), _p(out), N, H, W, st), "cs_op_t_mask")
    return out


def pair_ragged(wpacked, cout_pad, cin, k):
    """In place: the engine's load-time re-packing of the last chunk of a Cin % 32 == 16 layer (paired taps)."""
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.cs_op_pair_ragged(C.c_void_p(wpacked.data_ptr()), cout_pad, (cin + 31) // 32, k[0], k[1], k[2], st), "cs_op_pair_ragged")
    return wpacked


def packed_weight(w, cout_pad, device):
    return torch.from_numpy(pack.pack_conv(w.numpy() if isinstance(w, torch.Tensor) else w, cout_pad)).to(device)


def grid_sample(inp_hwdc, grid):
    lib = _lib.load()
    N, H, W, D, Cc = inp_hwdc.shape
    out32 = torch.empty_like(inp_hwdc)
    out16 = torch.empty(inp_hwdc.shape, dtype=torch.float16, device=inp_hwdc.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.cs_op_grid_sample3d(_p(inp_hwdc), _p(grid), _p(out32), _p(out16), N, D, H, W, st), "cs_op_grid_sample3d")
    return out32, out16


def chan_stats(x, eps=1e-5, with_partials=False):
    """x: [N, P, C] fp16/fp32 contiguous -> [N, C, 2] = (mean, 1/sqrt(var + eps)), biased variance
    (with_partials: also the first pass's partial sums [N, blocks, C, 2])."""
    lib = _lib.load()
    N, P, Cc = x.shape
    stats = torch.zeros(N, Cc, 2, dtype=torch.float32, device=x.device)
    part = torch.empty(lib.cs_op_chan_stats_partial_floats(N, P, Cc), dtype=torch.float32, device=x.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.cs_op_chan_stats(_p(x), int(x.dtype == torch.float32), N, P, Cc, eps, _p(part), _p(stats), st), "cs_op_chan_stats")
    return (stats, part.view(N, -1, Cc, 2)) if with_partials else stats


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
