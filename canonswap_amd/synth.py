"""Deterministic synthetic weights and inputs for the CanonSwap generator hot path.

No checkpoint ships with the reference (``pretrained_weights/`` is git-ignored), so every
parity / benchmark run in this repository uses weights produced here.  The generator is
framework independent (numpy PCG64 keyed by tensor name), so the oracle in the build container
and the HIP engine on the GPU box see bit-identical fp32 weights.

State-dict key names and shapes follow the reference modules exactly
(``src/modules/appearance_feature_extractor.py:17-36``, ``warping_network.py:15-44``,
``dense_motion.py:13-27``, ``spade_generator.py:13-39``, ``adaptive_modulate.py:73-126,485-520,
700-720``, ``util.py:80-344,515-527``) so they load with ``load_state_dict(strict=True)``.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

NUM_KP = 21
MODULES = ("appearance_feature_extractor", "warping_module", "spade_generator", "transfer", "refine")


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))


# weight families (precision-robustness tests): "uniform" = the default bounded init; "heavy_tail" = Student-t(3) weights of the
# same variance (a few weights per filter are 5-10 sigma, as in trained networks); see also rescale_feature_volume()
FAMILY = "uniform"


class _Builder:
    def __init__(self, seed: int, prefix: str):
        self.seed, self.prefix = seed, prefix
        self.sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def _r(self, key):
        return _rng(self.seed, self.prefix + "/" + key)

    def conv(self, name, cout, cin, *k, bias=True, gain=1.0, wname="weight"):
        fan_in = cin * int(np.prod(k)) if k else cin
        bound = gain * np.sqrt(3.0 / fan_in)  # unit-variance preserving for unit-variance input
        r = self._r(f"{name}.{wname}")
        if FAMILY == "heavy_tail":
            # Student-t with 3 degrees of freedom has variance 3: scale to the variance of U(-bound, bound) = bound^2 / 3
            w = r.standard_t(3, size=(cout, cin, *k)) * (bound / 3.0)
            w = np.clip(w, -12 * bound / np.sqrt(3.0), 12 * bound / np.sqrt(3.0))
            self.sd[f"{name}.{wname}"] = w.astype(np.float32)
        else:
            self.sd[f"{name}.{wname}"] = r.uniform(-bound, bound, size=(cout, cin, *k)).astype(np.float32)
        if bias:
            self.sd[f"{name}.bias"] = self._r(f"{name}.bias").uniform(-0.1, 0.1, size=(cout,)).astype(np.float32)

    def bn(self, name, c):
        self.sd[f"{name}.weight"] = self._r(f"{name}.weight").uniform(0.8, 1.2, size=(c,)).astype(np.float32)
        self.sd[f"{name}.bias"] = self._r(f"{name}.bias").uniform(-0.1, 0.1, size=(c,)).astype(np.float32)
        self.sd[f"{name}.running_mean"] = (0.1 * self._r(f"{name}.running_mean").standard_normal(c)).astype(np.float32)
        self.sd[f"{name}.running_var"] = self._r(f"{name}.running_var").uniform(0.6, 1.4, size=(c,)).astype(np.float32)
        self.sd[f"{name}.num_batches_tracked"] = np.array(100, dtype=np.int64)

    def gn(self, name, c):
        self.sd[f"{name}.weight"] = self._r(f"{name}.weight").uniform(0.8, 1.2, size=(c,)).astype(np.float32)
        self.sd[f"{name}.bias"] = self._r(f"{name}.bias").uniform(-0.1, 0.1, size=(c,)).astype(np.float32)

    def spectral_conv(self, name, cout, cin, *k, bias=True, gain=1.0):
        """Legacy ``torch.nn.utils.spectral_norm`` parametrisation: weight_orig / weight_u / weight_v
        with u, v converged by power iteration (default random u,v give sigma~0 and a saturated decoder,
        SURVEY.md section 7 step 2)."""
        self.conv(name, cout, cin, *k, bias=False, gain=gain, wname="weight_orig")
        if bias:
            self.sd[f"{name}.bias"] = self._r(f"{name}.bias").uniform(-0.1, 0.1, size=(cout,)).astype(np.float32)
            self.sd.move_to_end(f"{name}.weight_orig")
        w = self.sd[f"{name}.weight_orig"].reshape(cout, -1).astype(np.float64)
        u = self._r(f"{name}.weight_u").standard_normal(cout)
        u /= np.linalg.norm(u)
        for _ in range(200):
            v = w.T @ u
            v /= np.linalg.norm(v) + 1e-12
            u = w @ v
            u /= np.linalg.norm(u) + 1e-12
        self.sd[f"{name}.weight_u"] = u.astype(np.float32)
        self.sd[f"{name}.weight_v"] = v.astype(np.float32)

    def linear(self, name, cout, cin, gain=1.0):
        bound = gain * np.sqrt(3.0 / cin)
        self.sd[f"{name}.weight"] = self._r(f"{name}.weight").uniform(-bound, bound, size=(cout, cin)).astype(np.float32)
        self.sd[f"{name}.bias"] = self._r(f"{name}.bias").uniform(-0.1, 0.1, size=(cout,)).astype(np.float32)

    def resblock3d(self, name, c=32):
        # util.py:80-102 key order: conv1, conv2, norm1, norm2; small gain keeps the residual chain tame
        self.conv(f"{name}.conv1", c, c, 3, 3, 3, gain=1.0)
        self.conv(f"{name}.conv2", c, c, 3, 3, 3, gain=0.3)
        self.bn(f"{name}.norm1", c)
        self.bn(f"{name}.norm2", c)


def _appearance_feature_extractor(seed):
    b = _Builder(seed, "F")
    b.conv("first.conv", 64, 3, 3, 3); b.bn("first.norm", 64)
    b.conv("down_blocks.0.conv", 128, 64, 3, 3); b.bn("down_blocks.0.norm", 128)
    b.conv("down_blocks.1.conv", 256, 128, 3, 3); b.bn("down_blocks.1.norm", 256)
    b.conv("second", 512, 256, 1, 1)
    for i in range(6):
        b.resblock3d(f"resblocks_3d.3dr{i}")
    return b.sd


def _warping_module(seed):
    b = _Builder(seed, "W")
    p = "dense_motion_network"
    enc = [(110, 64), (64, 128), (128, 256), (256, 512), (512, 1024)]
    for i, (ci, co) in enumerate(enc):
        b.conv(f"{p}.hourglass.encoder.down_blocks.{i}.conv", co, ci, 3, 3, 3)
        b.bn(f"{p}.hourglass.encoder.down_blocks.{i}.norm", co)
    dec = [(1024, 512), (1024, 256), (512, 128), (256, 64), (128, 32)]
    for i, (ci, co) in enumerate(dec):
        b.conv(f"{p}.hourglass.decoder.up_blocks.{i}.conv", co, ci, 3, 3, 3)
        b.bn(f"{p}.hourglass.decoder.up_blocks.{i}.norm", co)
    b.conv(f"{p}.hourglass.decoder.conv", 142, 142, 3, 3, 3)
    b.bn(f"{p}.hourglass.decoder.norm", 142)
    b.conv(f"{p}.mask", 22, 142, 7, 7, 7, gain=2.0)
    b.conv(f"{p}.compress", 4, 32, 1, 1, 1, gain=4.0)
    b.bn(f"{p}.norm", 4)
    b.conv(f"{p}.occlusion", 1, 2272, 7, 7, gain=2.0)
    b.conv("third.conv", 256, 512, 3, 3); b.bn("third.norm", 256)
    b.conv("fourth", 256, 256, 1, 1)
    return b.sd


def _spade_block(b, name, fin, fout, label_nc=256):
    fmid = min(fin, fout)
    b.spectral_conv(f"{name}.conv_0", fmid, fin, 3, 3)
    b.spectral_conv(f"{name}.conv_1", fout, fmid, 3, 3)
    if fin != fout:
        b.spectral_conv(f"{name}.conv_s", fout, fin, 1, 1, bias=False)
    norms = [("norm_0", fin), ("norm_1", fmid)] + ([("norm_s", fin)] if fin != fout else [])
    for nn_, c in norms:
        b.conv(f"{name}.{nn_}.mlp_shared.0", 128, label_nc, 3, 3)
        b.conv(f"{name}.{nn_}.mlp_gamma", c, 128, 3, 3, gain=0.5)
        b.conv(f"{name}.{nn_}.mlp_beta", c, 128, 3, 3, gain=0.5)


def _spade_generator(seed):
    b = _Builder(seed, "G")
    b.conv("fc", 512, 256, 3, 3)
    for i in range(6):
        _spade_block(b, f"G_middle_{i}", 512, 512)
    _spade_block(b, "up_0", 512, 256)
    _spade_block(b, "up_1", 256, 64)
    b.conv("conv_img.0", 12, 64, 3, 3)
    return b.sd


def _transfer(seed):
    b = _Builder(seed, "T")
    for i in range(7):
        for cv in ("conv1", "conv2"):
            n = f"BottleNeck_2d.{i}.{cv}"
            b.conv(n, 512, 512, 3, 3, bias=False, gain=0.5 if cv == "conv2" else 1.0)
            b.sd[f"{n}.bias_param"] = b._r(f"{n}.bias_param").uniform(-0.1, 0.1, size=(512,)).astype(np.float32)
            b.linear(f"{n}.style_fc.0", 512, 512, gain=4.0)   # id is L2-normalised: |id_i| ~ 1/sqrt(512)
            b.linear(f"{n}.style_fc.2", 512, 512)
            b.conv(f"{n}.mask_conv.0", 1, 512, 3, 3, gain=2.0)
    for i in range(6):
        b.resblock3d(f"resblocks_3d.3dr{i}")
    return b.sd


def _refine(seed):
    b = _Builder(seed, "R")

    def stage3(name):
        b.conv(f"{name}.conv1", 32, 32, 3, 3, 3); b.gn(f"{name}.gn1", 32)
        b.conv(f"{name}.conv2", 32, 32, 3, 3, 3); b.gn(f"{name}.gn2", 32)

    for i in range(3):
        stage3(f"resblocks1.{i}")
    for i in range(3):
        n = f"resblocks2.{i}"
        b.conv(f"{n}.conv1", 512, 512, 3, 3)
        b.conv(f"{n}.conv2", 512, 512, 3, 3, gain=0.3)
        b.bn(f"{n}.norm1", 512); b.bn(f"{n}.norm2", 512)
    for i in range(3):
        stage3(f"resblocks3.{i}")
    return b.sd


_BUILDERS = {
    "appearance_feature_extractor": _appearance_feature_extractor,
    "warping_module": _warping_module,
    "spade_generator": _spade_generator,
    "transfer": _transfer,
    "refine": _refine,
}


def _motion_extractor(seed):
    """MotionExtractor(backbone='convnextv2_tiny', num_kp=21) state-dict (motion_extractor.py:18-35, convnextv2.py:48-108).
    Magnitudes are chosen so that the 18 residual blocks keep O(1) activations and every head output is non-degenerate."""
    b = _Builder(seed, "M")
    dims, depths = (96, 192, 384, 768), (3, 3, 9, 3)

    def ln(name, c):
        b.sd[f"{name}.weight"] = b._r(f"{name}.weight").uniform(0.8, 1.2, size=(c,)).astype(np.float32)
        b.sd[f"{name}.bias"] = b._r(f"{name}.bias").uniform(-0.1, 0.1, size=(c,)).astype(np.float32)

    b.conv("detector.downsample_layers.0.0", dims[0], 3, 4, 4)
    ln("detector.downsample_layers.0.1", dims[0])
    for i in range(3):
        ln(f"detector.downsample_layers.{i + 1}.0", dims[i])
        b.conv(f"detector.downsample_layers.{i + 1}.1", dims[i + 1], dims[i], 2, 2)
    for i, (c, n) in enumerate(zip(dims, depths)):
        for j in range(n):
            q = f"detector.stages.{i}.{j}"
            fan = 49
            bound = np.sqrt(3.0 / fan)
            b.sd[f"{q}.dwconv.weight"] = b._r(f"{q}.dwconv.weight").uniform(-bound, bound, size=(c, 1, 7, 7)).astype(np.float32)
            b.sd[f"{q}.dwconv.bias"] = b._r(f"{q}.dwconv.bias").uniform(-0.1, 0.1, size=(c,)).astype(np.float32)
            ln(f"{q}.norm", c)
            b.linear(f"{q}.pwconv1", 4 * c, c)
            b.sd[f"{q}.grn.gamma"] = (0.5 * b._r(f"{q}.grn.gamma").standard_normal((1, 1, 1, 4 * c))).astype(np.float32)
            b.sd[f"{q}.grn.beta"] = (0.1 * b._r(f"{q}.grn.beta").standard_normal((1, 1, 1, 4 * c))).astype(np.float32)
            b.linear(f"{q}.pwconv2", c, 4 * c, gain=0.5)
    ln("detector.norm", dims[-1])
    b.linear("detector.fc_kp", 3 * NUM_KP, dims[-1], gain=0.3)
    b.linear("detector.fc_scale", 1, dims[-1], gain=0.1)
    b.sd["detector.fc_scale.bias"] = np.array([1.1], np.float32)
    for k in ("pitch", "yaw", "roll"):
        b.linear(f"detector.fc_{k}", 66, dims[-1], gain=2.0)
    b.linear("detector.fc_t", 3, dims[-1], gain=0.1)
    b.linear("detector.fc_exp", 3 * NUM_KP, dims[-1], gain=0.05)
    return b.sd


_BUILDERS["motion_extractor"] = _motion_extractor


def make_state_dicts(seed: int = 0, modules=MODULES, family: str = "uniform") -> dict:
    """Return ``{module_name: OrderedDict[str, np.ndarray]}`` laid out like the reference's
    ``combined_weights.pth`` (``src/can_swap_e2e.py:87-100``).  The motion extractor is built on
    request: ``modules=MODULES + ("motion_extractor",)``.  family: "uniform" (default) or "heavy_tail"."""
    global FAMILY
    prev, FAMILY = FAMILY, family
    try:
        return {m: _BUILDERS[m](seed) for m in modules}
    finally:
        FAMILY = prev


def rescale_feature_volume(sds: dict, s: float) -> dict:
    """A weight family whose 32x16x64x64 feature volumes (f_s, f_can, f_swap, f_refined) are s times those of `sds` while the
    network function stays the same in exact arithmetic: every producer of the volume is scaled by s and every consumer's
    normalisation statistics / gates absorb it (BatchNorm running_mean * s, running_var * s^2; convs and biases that add to the
    residual stream * s; T's mask conv / s; GroupNorm affine * s).  What a trained checkpoint may do to the dynamic range of the
    fp16 tensors - used by the precision-robustness tests with s = 10, 0.1 and 100."""
    out = {m: OrderedDict((k, np.array(v, copy=True)) for k, v in sd.items()) for m, sd in sds.items()}
    s = float(s)

    def mul(sd, key, f):
        sd[key] = (sd[key].astype(np.float64) * f).astype(sd[key].dtype)

    def bn_in(sd, p):        # BatchNorm whose INPUT is scaled by s
        mul(sd, p + ".running_mean", s); mul(sd, p + ".running_var", s * s)

    def rb3d(sd, p):         # ResBlock3d: norm1 sees the stream, conv2 adds to it
        bn_in(sd, p + ".norm1"); mul(sd, p + ".conv2.weight", s); mul(sd, p + ".conv2.bias", s)

    F_ = out["appearance_feature_extractor"]
    mul(F_, "second.weight", s); mul(F_, "second.bias", s)
    for i in range(6):
        rb3d(F_, f"resblocks_3d.3dr{i}")
    W_ = out["warping_module"]
    bn_in(W_, "dense_motion_network.norm")      # BN after compress (conv bias is inside the BN input: scale it too)
    mul(W_, "dense_motion_network.compress.weight", 1.0)     # input * s -> output * s (linear)
    mul(W_, "dense_motion_network.compress.bias", s)
    mul(W_, "third.conv.bias", s); bn_in(W_, "third.norm")
    T_ = out["transfer"]
    for i in range(7):
        for cv in ("conv1", "conv2"):
            n = f"BottleNeck_2d.{i}.{cv}"
            mul(T_, n + ".bias_param", s)                     # out_mod + bias must scale with the stream
            mul(T_, n + ".mask_conv.0.weight", 1.0 / s)       # the gate keeps its value
    for i in range(6):
        rb3d(T_, f"resblocks_3d.3dr{i}")
    R_ = out["refine"]
    for blk in ("resblocks1", "resblocks3"):
        for i in range(3):
            mul(R_, f"{blk}.{i}.gn2.weight", s); mul(R_, f"{blk}.{i}.gn2.bias", s)    # GroupNorm output joins the stream
    for i in range(3):
        n = f"resblocks2.{i}"
        bn_in(R_, n + ".norm1"); mul(R_, n + ".conv2.weight", s); mul(R_, n + ".conv2.bias", s)
    return out


def to_torch(sds: dict) -> dict:
    import torch
    return {m: OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd.items())
            for m, sd in sds.items()}


# ------------------------------------------------------------------------------------------------
# synthetic per-frame inputs
# ------------------------------------------------------------------------------------------------
def _rot(pitch, yaw, roll):
    """Rotation convention of src/utils/camera.py:31-73 (degrees in, returns (Rz Ry Rx)^T)."""
    x, y, z = np.deg2rad([pitch, yaw, roll])
    rx = np.array([[1, 0, 0], [0, np.cos(x), -np.sin(x)], [0, np.sin(x), np.cos(x)]])
    ry = np.array([[np.cos(y), 0, np.sin(y)], [0, 1, 0], [-np.sin(y), 0, np.cos(y)]])
    rz = np.array([[np.cos(z), -np.sin(z), 0], [np.sin(z), np.cos(z), 0], [0, 0, 1]])
    return (rz @ ry @ rx).T


def make_frame_inputs(n_frames: int, seed: int = 1000, size: int = 256) -> dict:
    """Images U[0,1), key-points built with the transform of ``can_swap_e2e.py:228-256``:
    x_can = scale * kp ; x_t = scale * (kp @ R + exp) + t_xy."""
    img = np.empty((n_frames, 3, size, size), np.float32)
    x_t = np.empty((n_frames, NUM_KP, 3), np.float32)
    x_can = np.empty((n_frames, NUM_KP, 3), np.float32)
    for i in range(n_frames):
        r = _rng(seed + i, "frame")
        img[i] = r.random((3, size, size), dtype=np.float32)
        kp = np.clip(0.3 * r.standard_normal((NUM_KP, 3)), -1, 1)
        scale = r.uniform(0.9, 1.3)
        ang = r.uniform(-20, 20, size=3)
        exp = 0.02 * r.standard_normal((NUM_KP, 3))
        t = np.array([r.uniform(-0.1, 0.1), r.uniform(-0.1, 0.1), 0.0])
        x_can[i] = scale * kp
        x_t[i] = scale * (kp @ _rot(*ang) + exp) + t
    return {"img": img, "x_t": x_t, "x_can": x_can}


def make_smooth_images(n: int, seed: int = 2000, size: int = 256) -> np.ndarray:
    """(n,3,size,size) fp32 in [0,1]: low-frequency blobs + a little noise (inputs that make the motion extractor's pooled
    features differ between frames, which white noise does not)."""
    r = _rng(seed, "smooth")
    coarse = r.uniform(0, 1, size=(n, 3, 8, 8)).astype(np.float32)
    t = np.linspace(0, 7, size, dtype=np.float32)
    i0 = np.clip(np.floor(t).astype(np.int64), 0, 6)
    f = t - i0
    rows = coarse[:, :, i0, :] * (1 - f)[None, None, :, None] + coarse[:, :, i0 + 1, :] * f[None, None, :, None]
    img = rows[:, :, :, i0] * (1 - f) + rows[:, :, :, i0 + 1] * f
    img = img + 0.05 * r.standard_normal(img.shape).astype(np.float32)
    return np.clip(img, 0, 1).astype(np.float32)


def make_identity(seed: int = 7, n: int = 1) -> np.ndarray:
    v = _rng(seed, "identity").standard_normal((n, 512))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
