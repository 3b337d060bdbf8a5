"""CPU only: what Winograd F(2x2, 3x3) with fp16 operands costs G's image (pricing gate (a)).
G = SPADE decoder (spade_generator.py:41-59, util.py:305-344).  The oracle runs the frame in fp32; G is then re-run from the exact `seg` with
  direct   : every 3x3 conv of the listed layers on fp16-rounded inputs and weights, fp32 accumulate (what the engine's direct conv does)
  winograd : the same layers as 16 xi-GEMMs: V = B^T d B and U = G g G^T rounded to fp16 (V either from an fp32 transform with one rounding or
             from two fp16 add stages, as a packed-fp16 VALU transform would do), M = U V accumulated in fp32, Y = A^T M A in fp32
and the final image compared with the exact one.     python tests/diag/emul_winograd_G.py 63 3 19"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from canonswap_amd import synth  # noqa: E402
from oracle import canonswap_ref as O  # noqa: E402

torch.set_num_threads(16)
h = lambda t: t.half().float()
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G_ = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def wino_conv(x, w, bias, stage16=False, scale_v=1.0):
    """3x3 same conv as F(2x2,3x3) Winograd with fp16 V / U, fp32 accumulate."""
    n, c, H, W = x.shape
    k = w.shape[0]
    xp = F.pad(h(x), (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # n c th tw 4 4
    if stage16:                                                   # two fp16 add stages (rows, then columns)
        t = h(torch.einsum("ij,nchwjk->nchwik", BT, d))
        v = h(torch.einsum("nchwik,lk->nchwil", t, BT))
    else:
        v = h(torch.einsum("ij,nchwjk,lk->nchwil", BT, d, BT))
    u = h(torch.einsum("ij,kcjl,ml->kcim", G_, w, G_))           # k c 4 4
    m = torch.einsum("kcim,nchwim->nkhwim", u, v)
    y = torch.einsum("ai,nkhwim,bm->nkhwab", AT, m, AT)          # n k th tw 2 2
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, H, W)
    return y + bias.view(1, -1, 1, 1)


def direct16(x, w, bias):
    return F.conv2d(h(x), h(w), bias, padding=1)


def resblock(x, seg, sd, p, conv3):
    learned = (p + ".conv_s.weight_orig") in sd
    x_s = F.conv2d(O.spade(x, seg, sd, p + ".norm_s"), O.spectral_weight(sd, p + ".conv_s"), None) if learned else x
    dx = F.leaky_relu(O.spade(x, seg, sd, p + ".norm_0"), 0.2)
    dx = conv3(p + ".conv_0", dx, O.spectral_weight(sd, p + ".conv_0"), sd[p + ".conv_0.bias"])
    dx = F.leaky_relu(O.spade(dx, seg, sd, p + ".norm_1"), 0.2)
    dx = conv3(p + ".conv_1", dx, O.spectral_weight(sd, p + ".conv_1"), sd[p + ".conv_1.bias"])
    return x_s + dx


def decoder(sd, feature, conv3):
    seg = feature
    x = O.conv(feature, sd, "fc", 1)
    for i in range(6):
        x = resblock(x, seg, sd, f"G_middle_{i}", conv3)
    x = O.nearest_up(x, 1, 2, 2)
    x = resblock(x, seg, sd, "up_0", conv3)
    x = O.nearest_up(x, 1, 2, 2)
    x = resblock(x, seg, sd, "up_1", conv3)
    x = O.conv(F.leaky_relu(x, 0.2), sd, "conv_img.0", 1)
    return torch.sigmoid(O.pixel_shuffle2(x))


def main():
    sds = synth.to_torch(synth.make_state_dicts(0))
    inp = synth.make_frame_inputs(256, seed=1000, size=256)
    idv = torch.from_numpy(synth.make_identity(7))
    sdG = sds["spade_generator"]
    exact = lambda p, x, w, b: F.conv2d(x, w, b, padding=1)
    mid = lambda p: p.startswith("G_middle")
    wide = lambda p: p.startswith("G_middle") or p == "up_0.conv_0"
    rows = [("direct fp16, G_middle (engine today)", lambda p, x, w, b: direct16(x, w, b) if mid(p) else exact(p, x, w, b)),
            ("winograd fp16 (one rounding), G_middle", lambda p, x, w, b: wino_conv(x, w, b) if mid(p) else exact(p, x, w, b)),
            ("winograd fp16 (fp16 add stages), G_middle", lambda p, x, w, b: wino_conv(x, w, b, True) if mid(p) else exact(p, x, w, b)),
            ("direct fp16, all 3x3 of G", lambda p, x, w, b: direct16(x, w, b)),
            ("winograd G_middle + up_0.conv_0, direct fp16 rest", lambda p, x, w, b: wino_conv(x, w, b, True) if wide(p) else direct16(x, w, b)),
            ("winograd fp16 (add stages), all 3x3 of G", lambda p, x, w, b: wino_conv(x, w, b, True))]
    for j in [int(a) for a in sys.argv[1:]] or [63]:
        a = [torch.from_numpy(inp[k][j:j + 1]) for k in ("img", "x_t", "x_can")]
        with torch.no_grad():
            ref = O.swap_frame(sds, *a, idv, debug=False)
            chk = decoder(sdG, ref["seg"], exact)
            print(f"frame {j}: restated decoder vs oracle: max |diff| {float((chk - ref['out']).abs().max()):.2e}", flush=True)
            for name, fn in rows:
                t0 = time.time()
                out = decoder(sdG, ref["seg"], fn)
                mse = float(((out.double() - ref["out"].double()) ** 2).mean())
                print(f"frame {j} G {name:52s}: PSNR (rest exact) {O.psnr(out, ref['out']):6.2f} dB   mse {mse:.3e}   ({time.time() - t0:.1f}s)", flush=True)


if __name__ == "__main__":
    main()
