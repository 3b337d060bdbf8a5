"""ORACLE (test infrastructure, never the product path).

CPU fp32 restatement of the CanonSwap generator hot path, written functionally over the
reference's state-dict keys.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; the shipped engine
(``canonswap_amd``) never does and fails loudly when its HIP library is missing.

Every function cites the reference lines it restates (paths relative to the reference repo).
The arithmetic that the reference delegates to PyTorch ATen (conv, grid_sample, softmax ...)
is delegated to the same ATen CPU ops here; BatchNorm(eval), InstanceNorm, GroupNorm(32,32),
eval-mode spectral norm, PixelShuffle and nearest up-sampling are written out explicitly.

Pinning: the reference owns no golden vectors for this path, so the oracle is
pinned against outputs of the reference's own modules imported in the build container
(``tools/make_golden.py`` -> ``tests/golden/*.npz``, checked by ``tests/test_oracle_golden.py``).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

EPS_BN = 1e-5


def _id(t):
    return t


# ------------------------------------------------------------------------------------------------
# building blocks  (src/modules/util.py)
# ------------------------------------------------------------------------------------------------
def bn_eval(x, sd, p):
    """BatchNorm{2,3}d in eval mode: running statistics, eps 1e-5 (util.py:87-88,116-117 ...)."""
    shape = (1, -1) + (1,) * (x.dim() - 2)
    scale = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + EPS_BN)
    shift = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale.view(shape) + shift.view(shape)


def conv(x, sd, p, padding, weight=None):
    w = sd[p + ".weight"] if weight is None else weight
    b = sd.get(p + ".bias")
    fn = F.conv3d if w.dim() == 5 else F.conv2d
    return fn(x, w, b, padding=padding)


def same_block2d(x, sd, p, lrelu=False, q=_id):
    """util.py:193-211  conv -> BN -> ReLU | LeakyReLU(0.01)."""
    out = bn_eval(conv(x, sd, p + ".conv", 1), sd, p + ".norm")
    return q(F.leaky_relu(out, 0.01) if lrelu else F.relu(out))


def down_block2d(x, sd, p, q=_id):
    """util.py:150-166  conv -> BN -> ReLU -> AvgPool 2x2."""
    out = F.relu(bn_eval(conv(x, sd, p + ".conv", 1), sd, p + ".norm"))
    return q(F.avg_pool2d(q(out), 2))


def resblock3d(x, sd, p, q=_id):
    """util.py:94-102  pre-activation residual block, BN3d(eval) + ReLU."""
    out = q(F.relu(bn_eval(x, sd, p + ".norm1")))
    out = conv(out, sd, p + ".conv1", 1)
    out = q(F.relu(bn_eval(out, sd, p + ".norm2")))
    out = conv(out, sd, p + ".conv2", 1)
    return q(out + x)


def resblock2d(x, sd, p, slope=0.01, q=_id):
    """util.py:120-128  pre-activation residual block, BN2d(eval) + LeakyReLU(0.01)."""
    out = q(F.leaky_relu(bn_eval(x, sd, p + ".norm1"), slope))
    out = conv(out, sd, p + ".conv1", 1)
    out = q(F.leaky_relu(bn_eval(out, sd, p + ".norm2"), slope))
    out = conv(out, sd, p + ".conv2", 1)
    return q(out + x)


def down_block3d(x, sd, p, q=_id):
    """util.py:185-190  conv3d -> BN -> ReLU -> AvgPool(1,2,2)."""
    out = F.relu(bn_eval(conv(x, sd, p + ".conv", 1), sd, p + ".norm"))
    return q(F.avg_pool3d(q(out), (1, 2, 2)))


def nearest_up(x, fd, fh, fw):
    """F.interpolate(mode='nearest') with integer factors == index i -> i // f (util.py:143,297)."""
    if x.dim() == 5:
        return x.repeat_interleave(fd, 2).repeat_interleave(fh, 3).repeat_interleave(fw, 4)
    return x.repeat_interleave(fh, 2).repeat_interleave(fw, 3)


def up_block3d(x, sd, p, q=_id):
    """util.py:142-147  nearest x(1,2,2) -> conv3d -> BN -> ReLU."""
    out = nearest_up(x, 1, 2, 2)
    return q(F.relu(bn_eval(conv(out, sd, p + ".conv", 1), sd, p + ".norm")))


def group_norm_32_32(x, sd, p, eps=1e-5):
    """GroupNorm(32 groups, 32 channels) == per-(n,c) normalisation over D*H*W, biased variance
    (util.py:521-523)."""
    dims = tuple(range(2, x.dim()))
    mean = x.mean(dims, keepdim=True)
    var = x.var(dims, unbiased=False, keepdim=True)
    shape = (1, -1) + (1,) * (x.dim() - 2)
    return (x - mean) / torch.sqrt(var + eps) * sd[p + ".weight"].view(shape) + sd[p + ".bias"].view(shape)


def resblock3d_stage3_leak(x, sd, p, q=_id):
    """util.py:528-544 (shortcut is Identity for 32->32, upsample False)."""
    out = q(conv(x, sd, p + ".conv1", 1))
    out = q(F.leaky_relu(group_norm_32_32(out, sd, p + ".gn1"), 0.01))
    out = q(conv(out, sd, p + ".conv2", 1))
    out = group_norm_32_32(out, sd, p + ".gn2") + x
    return q(F.leaky_relu(out, 0.01))


# ------------------------------------------------------------------------------------------------
# F  (src/modules/appearance_feature_extractor.py:38-48)
# ------------------------------------------------------------------------------------------------
def appearance_feature_extractor(sd, img, q=_id):
    out = same_block2d(img, sd, "first", q=q)
    out = down_block2d(out, sd, "down_blocks.0", q=q)
    out = down_block2d(out, sd, "down_blocks.1", q=q)
    out = q(conv(out, sd, "second", 0))
    bs, c, h, w = out.shape
    f_s = out.view(bs, 32, 16, h, w)
    for i in range(6):
        f_s = resblock3d(f_s, sd, f"resblocks_3d.3dr{i}", q=q)
    return f_s


# ------------------------------------------------------------------------------------------------
# W  (src/modules/dense_motion.py, warping_network.py)
# ------------------------------------------------------------------------------------------------
def make_coordinate_grid(d, h, w, dtype=torch.float32):
    """util.py:41-58: x = 2*i/(w-1)-1 etc., last dim ordered (x, y, z)."""
    x = 2 * (torch.arange(w, dtype=dtype) / (w - 1)) - 1
    y = 2 * (torch.arange(h, dtype=dtype) / (h - 1)) - 1
    z = 2 * (torch.arange(d, dtype=dtype) / (d - 1)) - 1
    zz, yy, xx = torch.meshgrid(z, y, x, indexing="ij")
    return torch.stack([xx, yy, zz], -1)  # (d,h,w,3)


def kp2gaussian(kp, d, h, w, kp_variance=0.01):
    """util.py:17-38."""
    grid = make_coordinate_grid(d, h, w, kp.dtype).view(1, 1, d, h, w, 3)
    diff = grid - kp.view(kp.shape[0], kp.shape[1], 1, 1, 1, 3)
    return torch.exp(-0.5 * (diff ** 2).sum(-1) / kp_variance)


def grid_sample_3d_explicit(inp, grid):
    """Explicit restatement of F.grid_sample(5-D, bilinear==trilinear, zeros padding,
    align_corners=False) as called at warping_network.py:47 / dense_motion.py:50:
    ix = ((x+1)*W-1)/2 ..., 8 corners, out-of-range corners contribute 0."""
    n, c, d, h, w = inp.shape
    ix = ((grid[..., 0] + 1) * w - 1) / 2
    iy = ((grid[..., 1] + 1) * h - 1) / 2
    iz = ((grid[..., 2] + 1) * d - 1) / 2
    x0, y0, z0 = torch.floor(ix), torch.floor(iy), torch.floor(iz)
    out = torch.zeros((n, c) + grid.shape[1:4], dtype=inp.dtype)
    flat = inp.reshape(n, c, -1)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xc, yc, zc = x0 + dx, y0 + dy, z0 + dz
                wgt = (1 - (ix - xc).abs()) * (1 - (iy - yc).abs()) * (1 - (iz - zc).abs())
                ok = (xc >= 0) & (xc < w) & (yc >= 0) & (yc < h) & (zc >= 0) & (zc < d)
                idx = (zc.clamp(0, d - 1) * h + yc.clamp(0, h - 1)) * w + xc.clamp(0, w - 1)
                g = torch.gather(flat, 2, idx.long().view(n, 1, -1).expand(n, c, -1)).view(out.shape)
                out = out + g * (wgt * ok).unsqueeze(1)
    return out


def dense_motion(sd, feature, kp_driving, kp_source, q=_id):
    """dense_motion.py:67-104."""
    p = "dense_motion_network"
    bs, _, d, h, w = feature.shape
    K = kp_driving.shape[1]
    feat = q(F.relu(bn_eval(conv(feature, sd, p + ".compress", 0), sd, p + ".norm")))     # :70-72
    # create_sparse_motions :29-43
    ident = make_coordinate_grid(d, h, w).view(1, 1, d, h, w, 3)
    d2s = ident - kp_driving.view(bs, K, 1, 1, 1, 3) + kp_source.view(bs, K, 1, 1, 1, 3)
    sparse = torch.cat([ident.expand(bs, -1, -1, -1, -1, -1), d2s], 1)                   # (bs,K+1,d,h,w,3)
    # create_deformed_feature :45-53
    rep = feat.unsqueeze(1).expand(-1, K + 1, -1, -1, -1, -1).reshape(bs * (K + 1), -1, d, h, w)
    deformed = F.grid_sample(rep, sparse.reshape(bs * (K + 1), d, h, w, 3), align_corners=False)
    deformed = deformed.view(bs, K + 1, -1, d, h, w)
    # create_heatmap_representations :55-65
    heat = kp2gaussian(kp_driving, d, h, w) - kp2gaussian(kp_source, d, h, w)
    heat = torch.cat([torch.zeros(bs, 1, d, h, w), heat], 1).unsqueeze(2)
    inp = q(torch.cat([heat, deformed], 2).view(bs, -1, d, h, w))                            # :83-84
    # hourglass util.py:214-279
    outs = [inp]
    for i in range(5):
        outs.append(down_block3d(outs[-1], sd, f"{p}.hourglass.encoder.down_blocks.{i}", q=q))
    out = outs.pop()
    for i in range(5):
        out = up_block3d(out, sd, f"{p}.hourglass.decoder.up_blocks.{i}", q=q)
        out = torch.cat([out, outs.pop()], 1)
    pred = q(F.relu(bn_eval(conv(out, sd, p + ".hourglass.decoder.conv", 1), sd, p + ".hourglass.decoder.norm")))
    mask = F.softmax(conv(pred, sd, p + ".mask", 3), dim=1)                               # :88-89
    deformation = (sparse.permute(0, 1, 5, 2, 3, 4) * mask.unsqueeze(2)).sum(1).permute(0, 2, 3, 4, 1)  # :91-94
    occ = torch.sigmoid(conv(pred.reshape(bs, -1, h, w), sd, p + ".occlusion", 3))        # :98-102
    return {"mask": mask, "deformation": deformation, "occlusion_map": occ}


def warp(sd, feature_3d, kp_source, kp_driving, q=_id):
    """warping_network.py:49-62 (positional order: feature, kp_source, kp_driving)."""
    dm = dense_motion(sd, feature_3d, kp_driving=kp_driving, kp_source=kp_source, q=q)
    out = q(F.grid_sample(feature_3d, dm["deformation"], align_corners=False))
    return out, dm["occlusion_map"], dm


def warp_out(sd, out, occlusion_map=None, q=_id):
    """warping_network.py:64-71."""
    bs, c, d, h, w = out.shape
    out = same_block2d(out.reshape(bs, c * d, h, w), sd, "third", lrelu=True, q=q)
    out = conv(out, sd, "fourth", 0)
    if occlusion_map is not None:
        out = out * occlusion_map
    return q(out)


def warping_forward(sd, feature_3d, kp_driving, kp_source, q=_id):
    """warping_network.py:83-111 (keyword order: kp_driving, kp_source)."""
    out, occ, dm = warp(sd, feature_3d, kp_source=kp_source, kp_driving=kp_driving, q=q)
    return {"occlusion_map": occ, "deformation": dm["deformation"], "out": warp_out(sd, out, occ, q=q)}


# ------------------------------------------------------------------------------------------------
# T  (src/modules/adaptive_modulate.py:73-193, 310-349, 485-554)
# ------------------------------------------------------------------------------------------------
def style_vector(sd, p, latent):
    """style_fc: Linear -> LeakyReLU(0.2) -> Linear (adaptive_modulate.py:103-107,148)."""
    s = F.linear(latent, sd[p + ".style_fc.0.weight"], sd[p + ".style_fc.0.bias"])
    return F.linear(F.leaky_relu(s, 0.2), sd[p + ".style_fc.2.weight"], sd[p + ".style_fc.2.bias"])


def modulated_weight(sd, p, latent, eps=1e-8):
    """w_mod[n,o,i,ky,kx] = W[o,i,ky,kx]*style[n,i]; demodulate over (i,ky,kx) (:151-155)."""
    style = style_vector(sd, p, latent)
    w_mod = sd[p + ".weight"].unsqueeze(0) * style[:, None, :, None, None]
    return w_mod * torch.rsqrt((w_mod ** 2).sum(dim=(2, 3, 4), keepdim=True) + eps)


def adaptive_conv2d(x, sd, p, latent, q=_id):
    """adaptive_modulate.py:128-193 with use_learned_mask=True, bias=True, use_adaptive_norm=False."""
    n = x.shape[0]
    out_std = F.conv2d(x, sd[p + ".weight"], None, padding=1)
    w_mod = modulated_weight(sd, p, latent)
    out_mod = torch.cat([F.conv2d(x[i:i + 1], w_mod[i], None, padding=1) for i in range(n)], 0)  # groups=N
    out_mod = out_mod + sd[p + ".bias_param"].view(1, -1, 1, 1)
    mask = torch.sigmoid(F.conv2d(x, sd[p + ".mask_conv.0.weight"], sd[p + ".mask_conv.0.bias"], padding=1))
    return mask * out_mod + (1 - mask) * out_std, mask


def transfer(sd, x, dlatents, q=_id):
    """transfer_model2.forward (:522-554), return_mask=False."""
    bs, c, d, h, w = x.shape
    x = x.reshape(bs, c * d, h, w)
    for i in range(7):
        p = f"BottleNeck_2d.{i}"
        y, _ = adaptive_conv2d(x, sd, p + ".conv1", dlatents)          # ResnetBlock_Adaptive2D :337-349
        y = q(F.relu(y))
        y, _ = adaptive_conv2d(y, sd, p + ".conv2", dlatents)
        x = q(x + y)
    x = x.view(bs, c, d, h, w)
    for i in range(6):
        x = resblock3d(x, sd, f"resblocks_3d.3dr{i}", q=q)
    return x


# ------------------------------------------------------------------------------------------------
# R  (adaptive_modulate.py:700-733)
# ------------------------------------------------------------------------------------------------
def refine(sd, x, q=_id):
    for i in range(3):
        x = resblock3d_stage3_leak(x, sd, f"resblocks1.{i}", q=q)
    bs, c, d, h, w = x.shape
    x = x.reshape(bs, c * d, h, w)
    for i in range(3):
        x = resblock2d(x, sd, f"resblocks2.{i}", q=q)
    x = x.view(bs, c, d, h, w)
    for i in range(3):
        x = resblock3d_stage3_leak(x, sd, f"resblocks3.{i}", q=q)
    return x


# ------------------------------------------------------------------------------------------------
# G  (src/modules/spade_generator.py:41-59, util.py:282-344)
# ------------------------------------------------------------------------------------------------
def spectral_weight(sd, p):
    """Eval-mode legacy spectral_norm: W = W_orig / (u . (W_mat v)) (util.py:319-322)."""
    w = sd[p + ".weight_orig"]
    sigma = torch.dot(sd[p + ".weight_u"], torch.mv(w.reshape(w.shape[0], -1), sd[p + ".weight_v"]))
    return w / sigma


def instance_norm2d(x, eps=1e-5):
    """InstanceNorm2d(affine=False): per-(n,c) over H*W, biased variance (util.py:286)."""
    mean = x.mean((2, 3), keepdim=True)
    var = x.var((2, 3), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


def spade(x, seg, sd, p, q=_id):
    """util.py:295-302."""
    normalized = instance_norm2d(x)
    f = x.shape[2] // seg.shape[2]
    seg_r = nearest_up(seg, 1, f, f) if f > 1 else seg
    actv = q(F.relu(conv(seg_r, sd, p + ".mlp_shared.0", 1)))
    gamma = conv(actv, sd, p + ".mlp_gamma", 1)
    beta = conv(actv, sd, p + ".mlp_beta", 1)
    return normalized * (1 + gamma) + beta


def spade_resblock(x, seg, sd, p, q=_id):
    """util.py:329-344."""
    learned = (p + ".conv_s.weight_orig") in sd
    if learned:
        x_s = F.conv2d(q(spade(x, seg, sd, p + ".norm_s", q=q)), spectral_weight(sd, p + ".conv_s"), None)
        x_s = q(x_s)
    else:
        x_s = x
    dx = q(F.leaky_relu(spade(x, seg, sd, p + ".norm_0", q=q), 0.2))
    dx = q(F.conv2d(dx, spectral_weight(sd, p + ".conv_0"), sd[p + ".conv_0.bias"], padding=1))
    dx = q(F.leaky_relu(spade(dx, seg, sd, p + ".norm_1", q=q), 0.2))
    dx = F.conv2d(dx, spectral_weight(sd, p + ".conv_1"), sd[p + ".conv_1.bias"], padding=1)
    return q(x_s + dx)


def pixel_shuffle2(x):
    """nn.PixelShuffle(2): out[c, 2h+i, 2w+j] = in[c*4 + i*2 + j, h, w]."""
    n, c4, h, w = x.shape
    c = c4 // 4
    return x.view(n, c, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3).reshape(n, c, 2 * h, 2 * w)


def spade_decoder(sd, feature, q=_id):
    seg = feature
    x = q(conv(feature, sd, "fc", 1))
    for i in range(6):
        x = spade_resblock(x, seg, sd, f"G_middle_{i}", q=q)
    x = nearest_up(x, 1, 2, 2)
    x = spade_resblock(x, seg, sd, "up_0", q=q)
    x = nearest_up(x, 1, 2, 2)
    x = spade_resblock(x, seg, sd, "up_1", q=q)
    x = conv(q(F.leaky_relu(x, 0.2)), sd, "conv_img.0", 1)
    return torch.sigmoid(pixel_shuffle2(x))


# ------------------------------------------------------------------------------------------------
# wrappers  (src/can_swap_e2e.py, src/can_swap_pipeline_e2e.py:242-263)
# ------------------------------------------------------------------------------------------------
def conv_decode(sds, f, occ, q=_id):
    """can_swap_e2e.py:309-312."""
    return spade_decoder(sds["spade_generator"], warp_out(sds["warping_module"], f, occ, q=q), q=q)


def swap_frame(sds, img, x_t, x_can, source_id, debug=False, q=_id):
    """One frame (or batch of frames) through the per-frame loop body,
    can_swap_pipeline_e2e.py:242-263: F -> warp(x_t -> x_can) -> T -> R -> forward(x_can -> x_t) -> G."""
    with torch.no_grad():
        f_s = appearance_feature_extractor(sds["appearance_feature_extractor"], img, q=q)      # :242
        f_can, occ, _ = warp(sds["warping_module"], f_s, kp_source=x_t, kp_driving=x_can, q=q)  # :244
        res = {"f_s": f_s, "f_can": f_can, "occ": occ}
        if debug:
            res["rec_can"] = conv_decode(sds, f_can, occ, q=q)                                  # :248
        f_swap = transfer(sds["transfer"], f_can, source_id.expand(img.shape[0], -1), q=q)      # :253
        if debug:
            res["swap_can"] = conv_decode(sds, f_swap, occ, q=q)                                # :257
        f_ref = refine(sds["refine"], f_swap, q=q)                                              # :262
        ret = warping_forward(sds["warping_module"], f_ref, kp_driving=x_t, kp_source=x_can, q=q)  # :263
        out = spade_decoder(sds["spade_generator"], ret["out"], q=q)
        res.update(f_swap=f_swap, f_ref=f_ref, seg=ret["out"], deformation=ret["deformation"],
                   occ2=ret["occlusion_map"], out=out)
    return res


def prepare_source(img_u8: np.ndarray) -> torch.Tensor:
    """can_swap_e2e.py:126-145 (without the cv2 resize branch): u8 HWC -> fp32 1x3xHxW in [0,1]."""
    x = img_u8[np.newaxis] if img_u8.ndim == 3 else img_u8
    x = np.clip(x.astype(np.float32) / 255.0, 0, 1)
    return torch.from_numpy(x).permute(0, 3, 1, 2)


def prepare_videos(imgs) -> torch.Tensor:
    """can_swap_e2e.py:147-163: list of HxWx3 u8 -> Tx1x3xHxW fp32."""
    _imgs = np.array(imgs)[..., np.newaxis] if isinstance(imgs, list) else imgs
    y = np.clip(_imgs.astype(np.float32) / 255.0, 0, 1)
    return torch.from_numpy(y).permute(0, 4, 3, 1, 2)


def parse_output(out: torch.Tensor) -> np.ndarray:
    """can_swap_e2e.py:314-322: NCHW -> NHWC, clip, *255, clip, astype(uint8) (truncation)."""
    o = np.transpose(out.detach().cpu().numpy(), [0, 2, 3, 1])
    o = np.clip(o, 0, 1)
    return np.clip(o * 255, 0, 255).astype(np.uint8)


def headpose_pred_to_degree(pred):
    """src/utils/camera.py:14-28."""
    if pred.ndim > 1 and pred.shape[1] == 66:
        idx = torch.arange(66, dtype=torch.float32)
        return torch.sum(F.softmax(pred, dim=1) * idx, 1) * 3 - 97.5
    return pred


def get_rotation_matrix(pitch_, yaw_, roll_):
    """src/utils/camera.py:31-73 (degrees)."""
    x, y, z = [(a / 180 * np.pi).reshape(-1, 1) for a in (pitch_, yaw_, roll_)]
    bs = x.shape[0]
    one, zero = torch.ones(bs, 1), torch.zeros(bs, 1)
    rx = torch.cat([one, zero, zero, zero, torch.cos(x), -torch.sin(x), zero, torch.sin(x), torch.cos(x)], 1).view(bs, 3, 3)
    ry = torch.cat([torch.cos(y), zero, torch.sin(y), zero, one, zero, -torch.sin(y), zero, torch.cos(y)], 1).view(bs, 3, 3)
    rz = torch.cat([torch.cos(z), -torch.sin(z), zero, torch.sin(z), torch.cos(z), zero, zero, zero, one], 1).view(bs, 3, 3)
    return (rz @ ry @ rx).permute(0, 2, 1)


def transform_keypoint(kp_info: dict) -> torch.Tensor:
    """can_swap_e2e.py:228-256: s * (kp @ R + exp) + t_xy."""
    kp = kp_info["kp"]
    bs = kp.shape[0]
    num_kp = kp.shape[1] // 3 if kp.ndim == 2 else kp.shape[1]
    rot = get_rotation_matrix(headpose_pred_to_degree(kp_info["pitch"]), headpose_pred_to_degree(kp_info["yaw"]),
                              headpose_pred_to_degree(kp_info["roll"]))
    out = kp.reshape(bs, num_kp, 3) @ rot + kp_info["exp"].reshape(bs, num_kp, 3)
    out = out * kp_info["scale"][..., None]
    out[:, :, 0:2] = out[:, :, 0:2] + kp_info["t"][:, None, 0:2]
    return out


# ------------------------------------------------------------------------------------------------
# M  motion extractor: ConvNeXtV2-tiny + 7 linear heads
# (src/modules/convnextv2.py:15-144, motion_extractor.py:18-35)
# ------------------------------------------------------------------------------------------------
M_DIMS, M_DEPTHS = (96, 192, 384, 768), (3, 3, 9, 3)
M_HEADS = (("kp", 63), ("scale", 1), ("pitch", 66), ("yaw", 66), ("roll", 66), ("t", 3), ("exp", 63))   # convnextv2.py:98-106


def _ln_last(x, sd, p, eps=1e-6):
    """LayerNorm over the last (channel) axis, biased variance (util.py:388-396 both data formats)."""
    u = x.mean(-1, keepdim=True)
    v = ((x - u) ** 2).mean(-1, keepdim=True)
    return (x - u) / torch.sqrt(v + eps) * sd[p + ".weight"] + sd[p + ".bias"]


def convnext_block(x, sd, p):
    """convnextv2.py:34-46 on a channels-last tensor x (N,H,W,C): x + pw2(grn(gelu(pw1(ln(dw7x7(x))))))."""
    c = x.shape[-1]
    y = F.conv2d(x.permute(0, 3, 1, 2), sd[p + ".dwconv.weight"], sd[p + ".dwconv.bias"], padding=3, groups=c).permute(0, 2, 3, 1)
    y = _ln_last(y, sd, p + ".norm")
    y = F.gelu(y @ sd[p + ".pwconv1.weight"].t() + sd[p + ".pwconv1.bias"])            # exact (erf) GELU
    gx = torch.sqrt((y * y).sum(dim=(1, 2), keepdim=True))                              # util.py:365-368
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    y = sd[p + ".grn.gamma"] * (y * nx) + sd[p + ".grn.beta"] + y
    return x + (y @ sd[p + ".pwconv2.weight"].t() + sd[p + ".pwconv2.bias"])


def motion_features(sd, img):
    """forward_features (convnextv2.py:110-114); img (N,3,H,W) in [0,1] -> (N,768)."""
    p = "detector."
    x = F.conv2d(img, sd[p + "downsample_layers.0.0.weight"], sd[p + "downsample_layers.0.0.bias"], stride=4).permute(0, 2, 3, 1)
    x = _ln_last(x, sd, p + "downsample_layers.0.1")
    for i in range(4):
        if i > 0:
            x = _ln_last(x, sd, p + f"downsample_layers.{i}.0")
            x = F.conv2d(x.permute(0, 3, 1, 2), sd[p + f"downsample_layers.{i}.1.weight"], sd[p + f"downsample_layers.{i}.1.bias"],
                         stride=2).permute(0, 2, 3, 1)
        for j in range(M_DEPTHS[i]):
            x = convnext_block(x, sd, p + f"stages.{i}.{j}")
    return _ln_last(x.mean(dim=(1, 2)), sd, p + "norm")


def motion_extractor(sd, img):
    """MotionExtractor.forward (motion_extractor.py:33-35): dict of raw head outputs."""
    f = motion_features(sd, img)
    return {k: f @ sd[f"detector.fc_{k}.weight"].t() + sd[f"detector.fc_{k}.bias"] for k, _ in M_HEADS}


def get_kp_info(sd, img):
    """can_swap_e2e.py:174-199 with flag_refine_info=True."""
    info = motion_extractor(sd, img)
    bs = img.shape[0]
    for k in ("pitch", "yaw", "roll"):
        info[k] = headpose_pred_to_degree(info[k])[:, None]
    info["kp"] = info["kp"].reshape(bs, -1, 3)
    info["exp"] = info["exp"].reshape(bs, -1, 3)
    return info


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)
