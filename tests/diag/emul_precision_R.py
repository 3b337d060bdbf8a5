"""CPU only: which fp16 rounding inside R's twelve GroupNorm convs (util.py:528-544) costs the frame its PSNR - weights and conv inputs apart, the rest of
the frame exact fp32.  Prices a two-pass (weights-only) split against the engine's three-pass one.   python tests/diag/emul_precision_R.py 63"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from canonswap_amd import synth  # noqa: E402
from oracle import canonswap_ref as O  # noqa: E402

torch.set_num_threads(16)
sds = synth.to_torch(synth.make_state_dicts(0))
inp = synth.make_frame_inputs(256, seed=1000, size=256)
idv = torch.from_numpy(synth.make_identity(7))
h = lambda t: t.half().float()
ident = lambda t: t


def s3(x, sd, p, qa, qw):
    out = F.conv3d(qa(x), qw(sd[p + ".conv1.weight"]), sd.get(p + ".conv1.bias"), padding=1)
    out = F.leaky_relu(O.group_norm_32_32(out, sd, p + ".gn1"), 0.01)
    out = F.conv3d(qa(out), qw(sd[p + ".conv2.weight"]), sd.get(p + ".conv2.bias"), padding=1)
    return F.leaky_relu(O.group_norm_32_32(out, sd, p + ".gn2") + x, 0.01)


def refine(sd, x, qa=ident, qw=ident, q2=ident):
    for i in range(3):
        x = s3(x, sd, f"resblocks1.{i}", qa, qw)
    bs, c, d, hh, w = x.shape
    x = x.reshape(bs, c * d, hh, w)
    for i in range(3):
        x = O.resblock2d(x, sd, f"resblocks2.{i}", q=q2)
    x = x.view(bs, c, d, hh, w)
    for i in range(3):
        x = s3(x, sd, f"resblocks3.{i}", qa, qw)
    return x


for j in [int(a) for a in sys.argv[1:]] or [63]:
    a = [torch.from_numpy(inp[k][j:j + 1]) for k in ("img", "x_t", "x_can")]
    with torch.no_grad():
        ref = O.swap_frame(sds, *a, idv, debug=True)

        def final(f_ref):
            ret = O.warping_forward(sds["warping_module"], f_ref, kp_driving=a[1], kp_source=a[2])
            return O.psnr(O.spade_decoder(sds["spade_generator"], ret["out"]), ref["out"])
        rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
        for name, kw in [("exact", {}), ("GroupNorm convs: weights fp16", dict(qw=h)), ("GroupNorm convs: activations fp16", dict(qa=h)),
                         ("GroupNorm convs: both (plain fp16)", dict(qa=h, qw=h)), ("2-D convs' activations fp16 only", dict(q2=h))]:
            fr = refine(sds["refine"], ref["f_swap"], **kw)
            print(f"frame {j} R {name:36s}: rel {rel(fr, ref['f_ref']):.2e}  final PSNR with the rest exact {final(fr):6.2f} dB", flush=True)
