"""CPU only: what 1-D Winograd F(2,3) along W (fp16 V = d[j1] +- d[j2]: one rounding; fp16 U = G g along kw; fp32 accumulate; output
transform in fp32) costs the frame when T's 14 blend convs (adaptive_modulate.py:128-193) run on it, against today's direct fp16 convs.
The oracle runs the frame in fp32; T is re-run from the exact f_can with the named arithmetic in its 2-D part, the rest of the frame exact.
    python tests/diag/emul_winograd_T.py 63 3"""
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


def wino1d(x, w):
    """3x3 same conv, F(2,3) along W, direct along H."""
    n, c, H, W = x.shape
    xp = F.pad(h(x), (1, 1, 1, 1))
    d = xp.unfold(3, 4, 2)                                        # n c H+2 tw 4
    v = h(torch.einsum("ij,nchtj->nchti", BT, d))                 # one fp16 rounding of the sum of two fp16 values
    u = h(torch.einsum("ij,kcyj->kcyi", G_, w))                   # k c 3 4
    rows = v.unfold(2, 3, 1)                                      # n c H tw 4 3(ky)
    m = torch.einsum("kcyi,nchtiy->nkhti", u, rows)
    y = torch.einsum("ai,nkhti->nkhta", AT, m)                    # n k H tw 2
    return y.reshape(n, w.shape[0], H, W)


def direct16(x, w):
    return F.conv2d(h(x), h(w), None, padding=1)


def wino2d(x, w):
    n, c, H, W = x.shape
    xp = F.pad(h(x), (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    t = h(torch.einsum("ij,nchwjk->nchwik", BT, d))
    v = h(torch.einsum("nchwik,lk->nchwil", t, BT))
    u = h(torch.einsum("ij,kcjl,ml->kcim", G_, w, G_))
    m = torch.einsum("kcim,nchwim->nkhwim", u, v)
    y = torch.einsum("ai,nkhwim,bm->nkhwab", AT, m, AT)
    return y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], H, W)


def adaptive(x, sd, p, lat, conv):
    W = sd[p + ".weight"]
    out_std = conv(x, W)
    w_mod = O.modulated_weight(sd, p, lat)
    out_mod = conv(x, w_mod[0]) + sd[p + ".bias_param"].view(1, -1, 1, 1)
    mask = torch.sigmoid(F.conv2d(h(x), h(sd[p + ".mask_conv.0.weight"]), sd[p + ".mask_conv.0.bias"], padding=1))
    return mask * out_mod + (1 - mask) * out_std


def transfer(sd, x, lat, conv):
    bs, c, d, hh, w = x.shape
    x = x.reshape(bs, c * d, hh, w)
    for i in range(7):
        p = f"BottleNeck_2d.{i}"
        y = F.relu(adaptive(x, sd, p + ".conv1", lat, conv))
        y = adaptive(y, sd, p + ".conv2", lat, conv)
        x = x + y
    x = x.view(bs, c, d, hh, w)
    for i in range(6):
        x = O.resblock3d(x, sd, f"resblocks_3d.3dr{i}")
    return x


def main():
    sds = synth.to_torch(synth.make_state_dicts(0))
    inp = synth.make_frame_inputs(256, seed=1000, size=256)
    idv = torch.from_numpy(synth.make_identity(7))
    sdT = sds["transfer"]
    for j in [int(a) for a in sys.argv[1:]] or [63]:
        a = [torch.from_numpy(inp[k][j:j + 1]) for k in ("img", "x_t", "x_can")]
        with torch.no_grad():
            ref = O.swap_frame(sds, *a, idv, debug=False)

            def final(f_swap):
                f_ref = O.refine(sds["refine"], f_swap)
                ret = O.warping_forward(sds["warping_module"], f_ref, kp_driving=a[1], kp_source=a[2])
                out = O.spade_decoder(sds["spade_generator"], ret["out"])
                return O.psnr(out, ref["out"]), float(((out.double() - ref["out"].double()) ** 2).mean())
            rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
            for name, conv in [("direct fp16 (engine today)", direct16), ("F(2,3) along W, fp16 V / U", wino1d), ("F(2x2,3x3), fp16 add stages", wino2d)]:
                t0 = time.time()
                fs = transfer(sdT, ref["f_can"], idv, conv)
                ps, mse = final(fs)
                print(f"frame {j} T 2-D convs {name:30s}: rel {rel(fs, ref['f_swap']):.2e}  final PSNR (rest exact) {ps:6.2f} dB  mse {mse:.3e}  ({time.time() - t0:.1f}s)", flush=True)


if __name__ == "__main__":
    main()
