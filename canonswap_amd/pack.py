"""Load-time weight transformation: reference state-dicts -> blobs the gfx950 engine consumes.

Replaces ``can_swapper.load_cpk`` (src/can_swap_e2e.py:87-100): the same six state-dicts (same key names)
are ingested; eval-mode BatchNorm and the legacy spectral-norm parametrisation are folded into the
convolution weights, channels are reordered for the engine's channels-last layouts, and every
convolution is packed into the K-step order of the conv kernels (csrc/conv_halo_kernel.h):

    packed[kstep][row][kk] (fp16),  kstep = ((chunk*KD + kd)*KH + kh)*KW + kw,  in-channel = chunk*32 + kk

Feature volumes live as [N][H][W][D=16][C=32] on the device, so the reference's 512-channel 2-D view
(channel j = c*16 + d, e.g. warping_network.py:66) is the memory channel jm = d*32 + c.  ``MEM2REF[jm] = j``.
"""
from __future__ import annotations

import numpy as np

EPS_BN = 1e-5
MEM2REF = np.array([c * 16 + d for d in range(16) for c in range(32)], dtype=np.int64)


def bn_affine(sd, p):
    """Eval BatchNorm as y = x*s + t (running statistics, eps 1e-5)."""
    s = sd[p + ".weight"].astype(np.float64) / np.sqrt(sd[p + ".running_var"].astype(np.float64) + EPS_BN)
    t = sd[p + ".bias"].astype(np.float64) - sd[p + ".running_mean"].astype(np.float64) * s
    return s, t


def fold_conv_bn(w, b, s, t):
    """conv followed by y*s+t  ->  conv with w*s[o], b*s+t."""
    w = w.astype(np.float64) * s.reshape((-1,) + (1,) * (w.ndim - 1))
    b = (np.zeros(w.shape[0]) if b is None else b.astype(np.float64)) * s + t
    return w, b


def spectral_weight(sd, p):
    """Eval-mode torch.nn.utils.spectral_norm: W_orig / (u . (W_mat v)) (util.py:319-322)."""
    w = sd[p + ".weight_orig"].astype(np.float64)
    sigma = sd[p + ".weight_u"].astype(np.float64) @ (w.reshape(w.shape[0], -1) @ sd[p + ".weight_v"].astype(np.float64))
    return w / sigma


def pack_conv(w, cout_pad):
    """w: [Cout, Cin, (KD,) KH, KW] -> fp16 [nchunks*KD*KH*KW, cout_pad, 32]."""
    w = np.asarray(w, dtype=np.float64)
    if w.ndim == 4:
        w = w[:, :, None]
    co, ci, kd, kh, kw = w.shape
    nch = (ci + 31) // 32
    full = np.zeros((cout_pad, nch * 32, kd, kh, kw), np.float64)
    full[:co, :ci] = w
    full = full.reshape(cout_pad, nch, 32, kd, kh, kw).transpose(1, 3, 4, 5, 0, 2)
    return np.ascontiguousarray(full.reshape(nch * kd * kh * kw, cout_pad, 32).astype(np.float16))


def unpack_conv(packed, cout, cin, kd, kh, kw):
    """Inverse of pack_conv (tests)."""
    nch = (cin + 31) // 32
    cout_pad = packed.shape[1]
    full = packed.astype(np.float32).reshape(nch, kd, kh, kw, cout_pad, 32).transpose(4, 0, 5, 1, 2, 3)
    return full.reshape(cout_pad, nch * 32, kd, kh, kw)[:cout, :cin]


def interleave16(a, b):
    """Rows of a and b interleaved in blocks of 16: [a0..15, b0..15, a16..31, b16..31, ...]."""
    c = a.shape[0]
    assert c % 16 == 0 and a.shape == b.shape
    out = np.empty((2 * c,) + a.shape[1:], a.dtype)
    v = out.reshape((c // 16, 2, 16) + a.shape[1:])
    v[:, 0] = a.reshape((c // 16, 16) + a.shape[1:])
    v[:, 1] = b.reshape((c // 16, 16) + a.shape[1:])
    return out


def _f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def _pad(v, n):
    out = np.zeros(n, np.float64)
    out[: len(v)] = v
    return out


def _resblocks3d(out, prefix, sd):
    """6x ResBlock3d (util.py:80-102): conv1 carries norm2; norm1 of the next block rides on conv2's epilogue."""
    for i in range(6):
        p = f"resblocks_3d.3dr{i}"
        s2, t2 = bn_affine(sd, p + ".norm2")
        w1, b1 = fold_conv_bn(sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], s2, t2)
        out[f"{prefix}.rb{i}.c1.w"] = pack_conv(w1, 32)
        out[f"{prefix}.rb{i}.c1.b"] = _f32(b1)
        out[f"{prefix}.rb{i}.c2.w"] = pack_conv(sd[p + ".conv2.weight"], 32)
        out[f"{prefix}.rb{i}.c2.b"] = _f32(sd[p + ".conv2.bias"])
        if i < 5:
            s, t = bn_affine(sd, f"resblocks_3d.3dr{i + 1}.norm1")
            out[f"{prefix}.rb{i}.post.s"] = _f32(s)
            out[f"{prefix}.rb{i}.post.t"] = _f32(t)
    s, t = bn_affine(sd, "resblocks_3d.3dr0.norm1")   # pre-activation of block 0, per memory channel jm (c = jm % 32)
    out[f"{prefix}.pre0.s"] = _f32(np.tile(s, 16))
    out[f"{prefix}.pre0.t"] = _f32(np.tile(t, 16))


def _hi_lo(w):
    """[Cout][Cin][...] -> [Cout][2 Cin][...] = [W_hi | W_lo] (both fp16-representable): split-precision weights for a conv whose launch reads
    its input channels twice (engine.hip: wsplit_in), out = W_hi x + W_lo x.

    Accuracy: W_hi + W_lo == W to 2^-22 relative for |w| >= 2^-3.  Below that W_lo (<= 2^-11 |w|) is an fp16 SUBNORMAL, i.e. quantised at 2^-24
    absolute: W_hi + W_lo == W to 2^-25 absolute - for the |w| ~ 1e-2 of these layers about 2^-18 relative, still 7 bits beyond W_hi alone.  The
    scheme therefore relies on v_mfma_f32_16x16x32_f16 consuming fp16 denormal inputs exactly (it does not flush them, whatever the wave's
    denormal mode); tests/test_gpu_ops.py::test_conv_split_weights_use_fp16_subnormals holds that fact against a float64 convolution."""
    w = np.asarray(w, np.float64)
    hi = w.astype(np.float16).astype(np.float64)
    lo = (w - hi).astype(np.float16).astype(np.float64)
    return np.concatenate([hi, lo], axis=1)


def _pack_F(out, sd):
    s, t = bn_affine(sd, "first.norm")
    w, b = fold_conv_bn(sd["first.conv.weight"], sd["first.conv.bias"], s, t)
    out["F.first.w"] = _f32(w.reshape(64, 27))
    out["F.first.b"] = _f32(b)
    for i, (co) in enumerate((128, 256)):
        s, t = bn_affine(sd, f"down_blocks.{i}.norm")
        w, b = fold_conv_bn(sd[f"down_blocks.{i}.conv.weight"], sd[f"down_blocks.{i}.conv.bias"], s, t)
        out[f"F.down{i}.w"] = pack_conv(_hi_lo(w), co)
        out[f"F.down{i}.b"] = _f32(b)
    out["F.second.w"] = pack_conv(_hi_lo(sd["second.weight"][MEM2REF]), 512)
    out["F.second.b"] = _f32(sd["second.bias"][MEM2REF])
    _resblocks3d(out, "F", sd)


def _pack_W(out, sd):
    p = "dense_motion_network"
    s, t = bn_affine(sd, p + ".norm")
    w, b = fold_conv_bn(sd[p + ".compress.weight"], sd[p + ".compress.bias"], s, t)
    out["W.compress.w"] = _f32(w.reshape(4, 32))
    out["W.compress.b"] = _f32(b)
    for kind, blocks in (("enc", "encoder.down_blocks"), ("dec", "decoder.up_blocks")):
        for i in range(5):
            q = f"{p}.hourglass.{blocks}.{i}"
            s, t = bn_affine(sd, q + ".norm")
            w, b = fold_conv_bn(sd[q + ".conv.weight"], sd[q + ".conv.bias"], s, t)
            out[f"W.{kind}{i}.w"] = pack_conv(w, w.shape[0])
            out[f"W.{kind}{i}.b"] = _f32(b)
            if kind == "dec":                # the up-blocks run per output phase on the source grid (engine.hip)
                for (a, bb), (_, _, wab) in upsampled_conv3d_phases(w).items():
                    out[f"W.dec{i}.p{a}{bb}.w"] = pack_conv(wab, wab.shape[0])
    q = p + ".hourglass.decoder"
    s, t = bn_affine(sd, q + ".norm")
    w, b = fold_conv_bn(sd[q + ".conv.weight"], sd[q + ".conv.bias"], s, t)
    out["W.tail.w"] = pack_conv(w, 160)
    out["W.tail.b"] = _f32(_pad(b, 144))
    # 7x7x7 mask conv as a (7,7,1)-tap conv with (kw, c) output channels (summed over kw by dm_softmax_kernel):
    # w'[kw*22 + c][cin][kd][kh][0] = W[c][cin][kd][kh][kw]
    wm = sd[p + ".mask.weight"]                                       # [22][142][7][7][7]
    out["W.maskp.w"] = pack_conv(wm.transpose(4, 0, 1, 2, 3).reshape(154, 142, 7, 7, 1), 160)
    out["W.mask.b"] = _f32(_pad(sd[p + ".mask.bias"], 32))
    wo = sd[p + ".occlusion.weight"].reshape(142, 16, 7, 7)          # channel j = c*16 + d (dense_motion.py:100)
    # run on the MFMA conv as a 2-D (KH=7, KW=1) conv over grouped channels: input channel d*160 + c (c < 144 real) is depth
    # slice d, channel c; the 7 output channels are the horizontal taps kx (summed by occ_finish_kernel):
    #   w'[kx][d*160 + c][ky][0] = W_occ[0][c*16 + d][ky][kx]
    wg = np.zeros((7, 16, 160, 7, 1), np.float32)
    wg[:, :, :142, :, 0] = wo.transpose(3, 1, 0, 2)                  # [kx][d][c][ky]
    out["W.occp.w"] = pack_conv(wg.reshape(7, 16 * 160, 7, 1), 32)
    # second form (batched path): a 1x1 conv over the same grouped channels with the 49 taps as output channels,
    #   w49[ky*7 + kx][d*160 + c] = W_occ[0][c*16 + d][ky][kx]
    # - no halo, every activation read from LDS once for all taps; occ_finish49_kernel adds the 49 shifted partials
    w49 = np.zeros((49, 16, 160), np.float32)
    w49[:, :, :142] = wo.transpose(2, 3, 1, 0).reshape(49, 16, 142)   # [ky][kx][d][c]
    out["W.occ49.w"] = pack_conv(w49.reshape(49, 16 * 160, 1, 1), 64)
    out["W.occ.b"] = _f32(sd[p + ".occlusion.bias"].reshape(1))
    s, t = bn_affine(sd, "third.norm")
    w, b = fold_conv_bn(sd["third.conv.weight"][:, MEM2REF], sd["third.conv.bias"], s, t)
    out["W.third.w"] = pack_conv(w, 256)
    out["W.third.b"] = _f32(b)
    out["W.fourth.w"] = pack_conv(sd["fourth.weight"], 256)
    out["W.fourth.b"] = _f32(sd["fourth.bias"])


def _pack_T(out, sd):
    for i in range(7):
        for j in (1, 2):
            p = f"BottleNeck_2d.{i}.conv{j}"
            n = f"T.b{i}.c{j}"
            w = sd[p + ".weight"][MEM2REF][:, MEM2REF]                      # [o_m][i_m][3][3]
            out[n + ".w"] = pack_conv(interleave16(w, np.zeros_like(w)), 1024)   # rows: W | (w_mod filled by cs_set_identity)
            out[n + ".raw"] = _f32(w.transpose(0, 2, 3, 1).reshape(512, 9, 512))
            out[n + ".fc"] = _f32(np.concatenate([
                sd[p + ".style_fc.0.weight"].reshape(-1), sd[p + ".style_fc.0.bias"],
                sd[p + ".style_fc.2.weight"][MEM2REF].reshape(-1), sd[p + ".style_fc.2.bias"][MEM2REF]]))
            out[n + ".bias"] = _f32(sd[p + ".bias_param"][MEM2REF])
            out[n + ".mask.w"] = pack_conv(sd[p + ".mask_conv.0.weight"][:, MEM2REF], 16)
            out[n + ".mask.b"] = _f32(_pad(sd[p + ".mask_conv.0.bias"], 4))
    _resblocks3d(out, "T", sd)


def _pack_R(out, sd):
    for name, blk in (("s1", "resblocks1"), ("s3", "resblocks3")):
        for i in range(3):
            p, n = f"{blk}.{i}", f"R.{name}.{i}"
            for c in ("1", "2"):
                out[f"{n}.c{c}.w"] = pack_conv(sd[f"{p}.conv{c}.weight"], 32)
                # split-precision variant [W_hi | W_lo | W_hi] over the input channels, for inputs stored [hi | lo]: the
                # GroupNorm that follows each of these convs divides by the std of their output, and the fp16 rounding of
                # operands there is what limits the whole frame's PSNR
                w = np.asarray(sd[f"{p}.conv{c}.weight"], np.float32)
                hi = w.astype(np.float16).astype(np.float32)
                lo = (w - hi).astype(np.float16).astype(np.float32)
                out[f"{n}.c{c}.sp.w"] = pack_conv(np.concatenate([hi, lo, hi], axis=1), 32)    # chunks x [x_hi, x_hi, x_lo] (ConvParams::hilo)
                out[f"{n}.c{c}.b"] = _f32(sd[f"{p}.conv{c}.bias"])
                out[f"{n}.gn{c}.w"] = _f32(sd[f"{p}.gn{c}.weight"])
                out[f"{n}.gn{c}.b"] = _f32(sd[f"{p}.gn{c}.bias"])
    for i in range(3):
        p, n = f"resblocks2.{i}", f"R.rb2.{i}"
        s2, t2 = bn_affine(sd, p + ".norm2")
        w1, b1 = fold_conv_bn(sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], s2, t2)
        out[n + ".c1.w"] = pack_conv(w1[MEM2REF][:, MEM2REF], 512)
        out[n + ".c1.b"] = _f32(b1[MEM2REF])
        out[n + ".c2.w"] = pack_conv(sd[p + ".conv2.weight"][MEM2REF][:, MEM2REF], 512)
        out[n + ".c2.b"] = _f32(sd[p + ".conv2.bias"][MEM2REF])
        s1, t1 = bn_affine(sd, p + ".norm1")
        out[n + ".pre.s"] = _f32(s1[MEM2REF])
        out[n + ".pre.t"] = _f32(t1[MEM2REF])


def _pack_gb(out, n, sd, p):
    g, b = sd[p + ".mlp_gamma.weight"], sd[p + ".mlp_beta.weight"]
    c = g.shape[0]
    out[n + ".w"] = pack_conv(interleave16(g, b), ((2 * c + 127) // 128) * 128)
    out[n + ".bg"] = _f32(sd[p + ".mlp_gamma.bias"])
    out[n + ".bb"] = _f32(sd[p + ".mlp_beta.bias"])


def upsampled_conv_phases(w, s):
    """3x3 'same' conv applied to a nearest-x`s` up-sampled map (SPADE resizes seg to x's size, util.py:297-298) as convs on the
    SOURCE grid, one per output row phase a (y = s*i + a) and group of column phases that read the same source columns
    (s = 2: b = 0 | 1;  s = 4: b = 0 | 1,2 | 3): taps that land on the same source pixel add their weights, so a phase keeps
    only the 1-2 source rows / columns it reads (16 of 36 taps for s = 2, 36 of 144 for s = 4).  The column phases of a group
    are adjacent output pixels and become output-channel blocks.  Exact; zero padding of the up-sampled map is zero padding
    of the source grid.  Returns [(a, b0, nb, KH, PH, KW, PW, w [nb*Co][Ci][KH][KW])]; output channel = (b - b0)*Co + co."""
    w = np.asarray(w, np.float64)
    co, ci = w.shape[:2]
    off = lambda ph, d: (ph + d - 1) // s            # source offset (-1, 0, +1) read by tap d of output phase ph
    taps = lambda ph: tuple(sorted({off(ph, d) for d in range(3)}))
    groups = []                                      # runs of column phases with identical source-column sets
    for b in range(s):
        if groups and taps(groups[-1][0]) == taps(b):
            groups[-1].append(b)
        else:
            groups.append([b])
    res = []
    for a in range(s):
        rows = taps(a)
        for g in groups:
            cols = taps(g[0])
            wg = np.zeros((len(g) * co, ci, len(rows), len(cols)), np.float64)
            for k, b in enumerate(g):
                for dy in range(3):
                    for dx in range(3):
                        wg[k * co:(k + 1) * co, :, rows.index(off(a, dy)), cols.index(off(b, dx))] += w[:, :, dy, dx]
            res.append((a, g[0], len(g), len(rows), -rows[0], len(cols), -cols[0], wg))
    return res


def upsampled_conv3d_phases(w):
    """3x3x3 'same' conv after a nearest (1,2,2) up-sampling (UpBlock3d, util.py:142-147) as four convs on the SOURCE grid, one
    per output phase (a, b) of (y, x) = (2i + a, 2j + b): the three row / column taps collapse onto the two source rows /
    columns they read (12 of 27 taps).  Returns {(a, b): (PH, PW, w_ab [Co][Ci][3][2][2])}; zero padding carries over."""
    w = np.asarray(w, np.float64)
    off = lambda ph, d: (ph + d - 1) // 2
    res = {}
    for a in range(2):
        for b in range(2):
            rows = sorted({off(a, d) for d in range(3)}); cols = sorted({off(b, d) for d in range(3)})
            wab = np.zeros(w.shape[:3] + (2, 2), np.float64)
            for dy in range(3):
                for dx in range(3):
                    wab[:, :, :, rows.index(off(a, dy)), cols.index(off(b, dx))] += w[:, :, :, dy, dx]
            res[(a, b)] = (-rows[0], -cols[0], wab)
    return res


def _pack_G(out, sd):
    out["G.fc.w"] = pack_conv(sd["fc.weight"], 512)
    out["G.fc.b"] = _f32(sd["fc.bias"])
    groups = {
        "G.shared64": [f"G_middle_{b}.norm_{k}" for b in range(6) for k in (0, 1)],
        "G.shared128": ["up_0.norm_0", "up_0.norm_1", "up_0.norm_s"],
        "G.shared256": ["up_1.norm_0", "up_1.norm_1", "up_1.norm_s"],
    }
    for n, lst in groups.items():
        w = np.concatenate([sd[q + ".mlp_shared.0.weight"] for q in lst], 0)
        out[n + ".w"] = pack_conv(w, w.shape[0])
        out[n + ".b"] = _f32(np.concatenate([sd[q + ".mlp_shared.0.bias"] for q in lst]))
        s_up = {"G.shared128": 2, "G.shared256": 4}.get(n)
        if s_up:      # the same convs per output row phase on the 64x64 source grid (2.25x / 4x fewer taps, engine.hip run_G)
            for a, b0, nb, kh, ph, kw, pw, wg in upsampled_conv_phases(w, s_up):
                # x4 level, two-row phases (a = 0, 3): the column phases of the middle group read the same source column with the same summed
                # weights - one copy is packed, the engine's launch writes the value to both pixels (run_G); the one-row phase a = 1 keeps both
                # copies as output-channel blocks, its second output is taken by the duplicate ROW (a = 2 is never launched)
                if s_up == 4 and nb == 2 and a in (0, 3):
                    co = wg.shape[0] // nb
                    assert np.array_equal(wg[:co], wg[co:])
                    wg, nb = wg[:co], 1
                out[f"{n}.p{a}{b0}.w"] = pack_conv(wg, wg.shape[0])
                out[f"{n}.p{a}{b0}.b"] = _f32(np.tile(out[n + ".b"], nb))
    blocks = [(f"G.m{b}", f"G_middle_{b}") for b in range(6)] + [("G.up0", "up_0"), ("G.up1", "up_1")]
    for n, p in blocks:
        for k in ("0", "1"):
            _pack_gb(out, f"{n}.n{k}", sd, f"{p}.norm_{k}")
            w = spectral_weight(sd, f"{p}.conv_{k}")
            out[f"{n}.c{k}.w"] = pack_conv(w, w.shape[0])
            out[f"{n}.c{k}.b"] = _f32(sd[f"{p}.conv_{k}.bias"])
        if p + ".conv_s.weight_orig" in sd:
            # learned shortcut x_s = conv_s(norm_s(x, seg)) (util.py:329-344): no activation between SPADE's IN(x)(1 + gamma) + beta and the
            # bias-free 1x1 conv_s, so conv_s(beta) = conv3x3(actv; W_s W_beta) + W_s b_beta is ONE conv 128 -> fout (composed here in
            # float64) and only gamma is computed per input channel (engine.hip spade_shortcut)
            _pack_gb(out, f"{n}.ns", sd, f"{p}.norm_s")       # the fused gamma/beta form (CANONSWAP_SHORTCUT_ALGEBRA=0: the A/B knob)
            ws = spectral_weight(sd, f"{p}.conv_s").astype(np.float64)                  # [fout, fin, 1, 1]
            wg, wb = sd[f"{p}.norm_s.mlp_gamma.weight"], sd[f"{p}.norm_s.mlp_beta.weight"].astype(np.float64)     # [fin, 128, 3, 3]
            out[f"{n}.ng.w"] = pack_conv(wg, wg.shape[0])
            out[f"{n}.ng.b"] = _f32(sd[f"{p}.norm_s.mlp_gamma.bias"])
            wbs = np.einsum("oc,cjhw->ojhw", ws[:, :, 0, 0], wb)
            out[f"{n}.bs.w"] = pack_conv(wbs, wbs.shape[0])
            out[f"{n}.bs.b"] = _f32(ws[:, :, 0, 0] @ sd[f"{p}.norm_s.mlp_beta.bias"].astype(np.float64))
            out[f"{n}.cs.w"] = pack_conv(ws, ws.shape[0])
            out[f"{n}.cs.b"] = _f32(np.zeros(ws.shape[0]))
    out["G.img.w"] = pack_conv(sd["conv_img.0.weight"], 16)
    out["G.img.b"] = _f32(_pad(sd["conv_img.0.bias"], 16))


M_DIMS, M_DEPTHS = (96, 192, 384, 768), (3, 3, 9, 3)
M_HEADS = (("kp", 63), ("scale", 1), ("pitch", 66), ("yaw", 66), ("roll", 66), ("t", 3), ("exp", 63))


def split_precision_weight(w2d):
    """[Cout][Cin] fp32 -> [Cout][3 Cin] = [W_hi | W_hi | W_lo] (fp16-representable values); pairs with activations stored
    as [hi | lo | hi] (csrc/motion.hip) so that one fp16 MFMA conv yields W_hi v_hi + W_hi v_lo + W_lo v_hi."""
    w2d = np.asarray(w2d, np.float32)
    hi = w2d.astype(np.float16).astype(np.float32)
    lo = (w2d - hi).astype(np.float16).astype(np.float32)
    return np.concatenate([hi, hi, lo], axis=1)


def _pack_M(out, sd):
    """Motion extractor (convnextv2.py:48-108).  Linear layers become 1x1 convs; the 2x2 stride-2 downsample convs become
    1x1 convs over a space-to-depth input whose channel index is (dy*2+dx)*C + c; depth-wise weights go tap-major."""
    p = "detector."
    w = sd[p + "downsample_layers.0.0.weight"]                                    # [96][3][4][4] -> [k = ci*16+dy*4+dx][96]
    out["M.stem.w"] = _f32(w.reshape(96, 48).T)
    out["M.stem.b"] = _f32(sd[p + "downsample_layers.0.0.bias"])
    out["M.stem.ln.g"] = _f32(sd[p + "downsample_layers.0.1.weight"])
    out["M.stem.ln.b"] = _f32(sd[p + "downsample_layers.0.1.bias"])
    for i, (c, n) in enumerate(zip(M_DIMS, M_DEPTHS)):
        for j in range(n):
            q, o = f"{p}stages.{i}.{j}", f"M.s{i}.{j}"
            out[o + ".dw.w"] = _f32(sd[q + ".dwconv.weight"].reshape(c, 49).T)
            out[o + ".dw.b"] = _f32(sd[q + ".dwconv.bias"])
            out[o + ".ln.g"] = _f32(sd[q + ".norm.weight"]); out[o + ".ln.b"] = _f32(sd[q + ".norm.bias"])
            out[o + ".grn.g"] = _f32(sd[q + ".grn.gamma"].reshape(-1)); out[o + ".grn.b"] = _f32(sd[q + ".grn.beta"].reshape(-1))
            out[o + ".pw1.w"] = pack_conv(split_precision_weight(sd[q + ".pwconv1.weight"])[:, :, None, None], 4 * c)
            out[o + ".pw1.b"] = _f32(sd[q + ".pwconv1.bias"])
            out[o + ".pw2.w"] = pack_conv(split_precision_weight(sd[q + ".pwconv2.weight"])[:, :, None, None], c if c % 64 == 0 else -(-c // 128) * 128)   # stage 0: one 128-channel block (engine.hip)
            out[o + ".pw2.b"] = _f32(sd[q + ".pwconv2.bias"])
        if i < 3:
            q, o = f"{p}downsample_layers.{i + 1}", f"M.ds{i}"
            out[o + ".ln.g"] = _f32(sd[q + ".0.weight"]); out[o + ".ln.b"] = _f32(sd[q + ".0.bias"])
            w = sd[q + ".1.weight"]                                               # [2C][C][2][2] -> [2C][(dy,dx,c)]
            out[o + ".w"] = pack_conv(split_precision_weight(np.transpose(w, (0, 2, 3, 1)).reshape(2 * c, 4 * c))[:, :, None, None], 2 * c)
            out[o + ".b"] = _f32(sd[q + ".1.bias"])
    out["M.norm.g"] = _f32(sd[p + "norm.weight"]); out["M.norm.b"] = _f32(sd[p + "norm.bias"])
    out["M.head.w"] = _f32(np.concatenate([sd[f"{p}fc_{k}.weight"] for k, _ in M_HEADS], 0))
    out["M.head.b"] = _f32(np.concatenate([sd[f"{p}fc_{k}.bias"] for k, _ in M_HEADS], 0))
    assert out["M.head.w"].shape == (328, 768)


def _np_sd(sd):
    return {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in sd.items()}


def build_blobs(state_dicts: dict) -> dict:
    """``state_dicts``: {'appearance_feature_extractor', 'warping_module', 'spade_generator', 'transfer',
    'refine'[, 'motion_extractor']} -> {blob name: contiguous ndarray}; values may be torch tensors or numpy arrays."""
    out: dict = {}
    _pack_F(out, _np_sd(state_dicts["appearance_feature_extractor"]))
    _pack_W(out, _np_sd(state_dicts["warping_module"]))
    _pack_T(out, _np_sd(state_dicts["transfer"]))
    _pack_R(out, _np_sd(state_dicts["refine"]))
    _pack_G(out, _np_sd(state_dicts["spade_generator"]))
    if "motion_extractor" in state_dicts:                      # optional
        _pack_M(out, _np_sd(state_dicts["motion_extractor"]))
    return out
