"""CPU emulation (no GPU): which operand split do M's pointwise GEMMs need?  The motion extractor's linear layers run on the fp16 MFMA
with fp32 accumulation; variants of the operand representation against the fp32 oracle on smooth frames:
    v3  : W_hi x_hi + W_hi x_lo + W_lo x_hi      (round 3 - 5: three MFMA passes, activations stored as [hi | lo | hi])
    v2w : (W_hi + W_lo) x_hi                     (weights exact to 2^-22, activations rounded to fp16: two passes, activations 2 bytes)
    v2a : W_hi (x_hi + x_lo)                     (activations exact, weights rounded: two passes)
    v1  : W_hi x_hi
Prints max |d| of kp / exp / t / scale and of the transformed key-points.  python tests/diag/emul_m_split.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from canonswap_amd import synth
from oracle import canonswap_ref as O

sd = synth.to_torch(synth.make_state_dicts(0, modules=("motion_extractor",)))["motion_extractor"]
imgs = torch.from_numpy(synth.make_smooth_images(6, seed=2000, size=256))


def hi(t):
    return t.half().float()


def lin(x, W, b, variant):
    if variant == "fp32":
        return x @ W.t() + b
    xh, Wh = hi(x), hi(W)
    xl, Wl = hi(x - xh), hi(W - Wh)
    if variant == "v3":
        return (xh @ Wh.t() + xl @ Wh.t() + xh @ Wl.t()) + b
    if variant == "v2w":
        return (xh @ Wh.t() + xh @ Wl.t()) + b
    if variant == "v2a":
        return (xh @ Wh.t() + xl @ Wh.t()) + b
    return xh @ Wh.t() + b


def block(x, p, v):
    c = x.shape[-1]
    y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), sd[p + ".dwconv.weight"], sd[p + ".dwconv.bias"], padding=3, groups=c).permute(0, 2, 3, 1)
    y = O._ln_last(y, sd, p + ".norm")
    v1_, v2_, vd_ = (v.split("/") + [v, v])[:3] if "/" in v else (v, v, v)
    y = torch.nn.functional.gelu(lin(y, sd[p + ".pwconv1.weight"], sd[p + ".pwconv1.bias"], v1_))
    gx = torch.sqrt((y * y).sum(dim=(1, 2), keepdim=True))
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    y = sd[p + ".grn.gamma"] * (y * nx) + sd[p + ".grn.beta"] + y
    return x + lin(y, sd[p + ".pwconv2.weight"], sd[p + ".pwconv2.bias"], v2_)


def run(v):
    p = "detector."
    x = torch.nn.functional.conv2d(imgs, sd[p + "downsample_layers.0.0.weight"], sd[p + "downsample_layers.0.0.bias"], stride=4).permute(0, 2, 3, 1)
    x = O._ln_last(x, sd, p + "downsample_layers.0.1")
    for i in range(4):
        if i > 0:
            x = O._ln_last(x, sd, p + f"downsample_layers.{i}.0")
            N, H, W, C = x.shape
            w = sd[p + f"downsample_layers.{i}.1.weight"]                    # (2C, C, 2, 2) -> linear over (dy, dx, c)
            xs = x.reshape(N, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, H // 2, W // 2, 4 * C)
            x = lin(xs, w.permute(0, 2, 3, 1).reshape(w.shape[0], 4 * C), sd[p + f"downsample_layers.{i}.1.bias"], v.split("/")[2] if "/" in v else v)
        for j in range(O.M_DEPTHS[i]):
            x = block(x, p + f"stages.{i}.{j}", v)
    f = O._ln_last(x.mean(dim=(1, 2)), sd, p + "norm")
    info = {k: f @ sd[f"detector.fc_{k}.weight"].t() + sd[f"detector.fc_{k}.bias"] for k, _ in O.M_HEADS}
    bs = imgs.shape[0]
    out = dict(info)
    for k in ("pitch", "yaw", "roll"):
        out[k] = O.headpose_pred_to_degree(info[k])[:, None]
    out["kp"] = info["kp"].reshape(bs, -1, 3); out["exp"] = info["exp"].reshape(bs, -1, 3)
    return out, O.transform_keypoint(out)


with torch.no_grad():
    ref, xr = run("fp32")
    chk, _ = O.get_kp_info(sd, imgs), None
    assert (chk["kp"] - ref["kp"]).abs().max() < 1e-5
    for v in ("v3", "v2w", "v2a", "v1", "v3/v2w/v3", "v2w/v3/v3", "v2w/v2w/v3", "v3/v2w/v2w"):
        got, x = run(v)
        d = {k: (got[k] - ref[k]).abs().max().item() for k in ("kp", "exp", "t", "scale", "pitch", "yaw", "roll")}
        print(v, " ".join(f"{k} {e:.2e}" for k, e in d.items()), f"| x_t {(x - xr).abs().max().item():.2e}")
