"""CPU only: which fp16 rounding inside F (appearance_feature_extractor.py:38-48) costs the frame its PSNR - per conv, weights and conv inputs apart; the rest of
the frame is the exact fp32 oracle.  This is the measurement behind the split-precision weights of F's three 2-D convs (engine.hip: wsplit_in).
    python tests/diag/emul_precision_F.py 63"""
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
def convq(x, sd, p, pad, qa, qw):
    w = sd[p + ".weight"]; fn = F.conv3d if w.dim() == 5 else F.conv2d
    return fn(qa(x), qw(w), sd.get(p + ".bias"), padding=pad)
def Fx(sd, img, q1=ident, q2=ident, qs=ident, q3=ident, qw2=ident, qw3=ident, qwa=None, qwb=None, qwc=None, qa_b=None, qa_c=None):
    qwa = qwa or qw2; qwb = qwb or qw2; qwc = qwc or qw2; qa_b = qa_b or q2; qa_c = qa_c or q2
    # q1: activation rounding into down0 (first conv's output), q2: into down1 / second, qs: the fp32 residual stream start, q3: resblock conv inputs
    out = F.relu(O.bn_eval(convq(img, sd, "first.conv", 3 if sd["first.conv.weight"].shape[-1] == 7 else 1, ident, ident), sd, "first.norm"))
    out = F.relu(O.bn_eval(convq(out, sd, "down_blocks.0.conv", 1, q1, qwa), sd, "down_blocks.0.norm")); out = F.avg_pool2d(out, 2)
    out = F.relu(O.bn_eval(convq(out, sd, "down_blocks.1.conv", 1, qa_b, qwb), sd, "down_blocks.1.norm")); out = F.avg_pool2d(out, 2)
    out = qs(convq(out, sd, "second", 0, qa_c, qwc))
    bs, c, hh, w = out.shape
    x = out.view(bs, 32, 16, hh, w)
    for i in range(6):
        p = f"resblocks_3d.3dr{i}"
        o = convq(F.relu(O.bn_eval(x, sd, p + ".norm1")), sd, p + ".conv1", 1, q3, qw3)
        o = convq(F.relu(O.bn_eval(o, sd, p + ".norm2")), sd, p + ".conv2", 1, q3, qw3)
        x = o + x
    return x
for j in [int(a) for a in sys.argv[1:]] or [63]:
    a = [torch.from_numpy(inp[k][j:j + 1]) for k in ("img", "x_t", "x_can")]
    with torch.no_grad():
        ref = O.swap_frame(sds, *a, idv, debug=True)
        def final(f_s):
            f_can, occ, _ = O.warp(sds["warping_module"], f_s, kp_source=a[1], kp_driving=a[2])
            f_swap = O.transfer(sds["transfer"], f_can, idv)
            f_ref = O.refine(sds["refine"], f_swap)
            ret = O.warping_forward(sds["warping_module"], f_ref, kp_driving=a[1], kp_source=a[2])
            return O.psnr(O.spade_decoder(sds["spade_generator"], ret["out"]), ref["out"])
        rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
        sdF = sds["appearance_feature_extractor"]
        print("check exact:", rel(Fx(sdF, a[0]), ref["f_s"]))
        for name, kw in [("weights down0", dict(qwa=h)), ("weights down1", dict(qwb=h)), ("weights second", dict(qwc=h)),
                         ("act into down1", dict(qa_b=h)), ("act into second", dict(qa_c=h)), ("second: w + act", dict(qwc=h, qa_c=h)), ("down1+second: w + act", dict(qwb=h, qwc=h, qa_b=h, qa_c=h))]:
            fs = Fx(sdF, a[0], **kw)
            print(f"frame {j} F {name:26s}: rel {rel(fs, ref['f_s']):.2e}  final PSNR with the rest exact {final(fs):6.2f} dB", flush=True)
