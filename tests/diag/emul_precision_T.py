"""CPU only: which fp16 rounding inside T (adaptive_modulate.py:128-193, 522-554) costs the frame its PSNR.  The oracle runs the whole frame in fp32 except for
the roundings named per row (weights / conv inputs of the 14 blend convs and of the six ResBlock3d); the final image is compared with the exact one.
    python tests/diag/emul_precision_T.py 63"""
import sys, time, torch, torch.nn.functional as F
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from canonswap_amd import synth
from oracle import canonswap_ref as O
torch.set_num_threads(16)
sds = synth.to_torch(synth.make_state_dicts(0))
inp = synth.make_frame_inputs(256, seed=1000, size=256)
idv = torch.from_numpy(synth.make_identity(7))
h = lambda t: t.half().float()
ident = lambda t: t

def adaptive(x, sd, p, lat, qa, qw, qm):
    xin = qa(x)
    W = sd[p + ".weight"]
    out_std = F.conv2d(xin, qw(W), None, padding=1)
    w_mod = O.modulated_weight(sd, p, lat)
    out_mod = F.conv2d(xin, qw(w_mod[0]), None, padding=1) + sd[p + ".bias_param"].view(1, -1, 1, 1)
    mask = torch.sigmoid(F.conv2d(qm(x), qm(sd[p + ".mask_conv.0.weight"]), sd[p + ".mask_conv.0.bias"], padding=1))
    return mask * out_mod + (1 - mask) * out_std

def rb3(x, sd, p, qa, qw):
    out = qa(F.relu(O.bn_eval(x, sd, p + ".norm1")))
    out = F.conv3d(out, qw(sd[p + ".conv1.weight"]), sd.get(p + ".conv1.bias"), padding=1)
    out = qa(F.relu(O.bn_eval(out, sd, p + ".norm2")))
    out = F.conv3d(out, qw(sd[p + ".conv2.weight"]), sd.get(p + ".conv2.bias"), padding=1)
    return out + x

def transfer(sd, x, lat, qa=ident, qw=ident, qm=ident, qa3=None, qw3=None):
    qa3 = qa if qa3 is None else qa3; qw3 = qw if qw3 is None else qw3
    bs, c, d, hh, w = x.shape
    x = x.reshape(bs, c * d, hh, w)
    for i in range(7):
        p = f"BottleNeck_2d.{i}"
        y = F.relu(adaptive(x, sd, p + ".conv1", lat, qa, qw, qm))
        y = adaptive(y, sd, p + ".conv2", lat, qa, qw, qm)
        x = x + y
    x = x.view(bs, c, d, hh, w)
    for i in range(6):
        x = rb3(x, sd, f"resblocks_3d.3dr{i}", qa3, qw3)
    return x

for j in [int(a) for a in sys.argv[1:]] or [63]:
    a = [torch.from_numpy(inp[k][j:j + 1]) for k in ("img", "x_t", "x_can")]
    with torch.no_grad():
        ref = O.swap_frame(sds, *a, idv, debug=True)
        def final(f_swap):
            f_ref = O.refine(sds["refine"], f_swap)
            ret = O.warping_forward(sds["warping_module"], f_ref, kp_driving=a[1], kp_source=a[2])
            return O.psnr(O.spade_decoder(sds["spade_generator"], ret["out"]), ref["out"])
        rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
        sdT = sds["transfer"]
        for name, kw in [("exact", {}), ("weights fp16", dict(qw=h)), ("activations fp16", dict(qa=h, qm=h)), ("both (engine-like)", dict(qa=h, qw=h, qm=h)),
                         ("both, 2-D part only", dict(qa=h, qw=h, qm=h, qa3=ident, qw3=ident)), ("both, 3-D resblocks only", dict(qa3=h, qw3=h))]:
            t0 = time.time()
            fs = transfer(sdT, ref["f_can"], idv, **kw)
            print(f"frame {j} T {name:28s}: rel {rel(fs, ref['f_swap']):.2e}  final PSNR with the rest exact {final(fs):6.2f} dB  ({time.time()-t0:.1f}s)", flush=True)
