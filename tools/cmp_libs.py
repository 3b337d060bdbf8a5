"""GPU box: the same frames through two builds of the library, compared bit for bit (stage outputs of the whole loop body, default and latency mode,
and the motion extractor's raw head outputs):
    python tools/cmp_libs.py tools/bin/base.so ""        ("" = the shipped library)
Each build runs in its own process (the library is chosen at import: CANONSWAP_LIB)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from canonswap_amd import synth
from canonswap_amd.can_swap_e2e import can_swapper
B = 6
sds = synth.to_torch(synth.make_state_dicts(0))
out = {}
for lat in (False, True):
    sw = can_swapper(None, state_dicts=sds, max_batch=B, latency_mode=lat)
    inp = synth.make_frame_inputs(B, seed=1000, size=256)
    idv = torch.from_numpy(synth.make_identity(7)).cuda()
    args = [torch.from_numpy(inp[k]).cuda() for k in ("img", "x_t", "x_can")]
    nb = 1 if lat else B
    r = sw.swap_frames(*(a[:nb] for a in args), idv, debug=True)
    for k, v in r.items():
        if torch.is_tensor(v):
            out[("lat." if lat else "") + k] = v.cpu()
sdm = synth.to_torch(synth.make_state_dicts(0, modules=synth.MODULES + ("motion_extractor",)))
swm = can_swapper(None, state_dicts=sdm, max_batch=5)
out["M.raw"] = swm.engine.motion_extract_raw(torch.from_numpy(synth.make_smooth_images(5, seed=77, size=256)).cuda()).cpu()
torch.save(out, sys.argv[1])
''' % ROOT


def run(lib, path):
    env = dict(os.environ)
    if lib:
        env["CANONSWAP_LIB"] = lib
    else:
        env.pop("CANONSWAP_LIB", None)
    subprocess.run([sys.executable, "-c", CHILD, path], check=True, env=env, cwd=ROOT)


def main():
    import torch
    a, b = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as d:
        pa, pb = os.path.join(d, "a.pt"), os.path.join(d, "b.pt")
        run(a, pa); run(b, pb)
        A, Bv = torch.load(pa), torch.load(pb)
    bad = 0
    for k in A:
        same = torch.equal(A[k], Bv[k])
        bad += not same
        print("%-24s %s %s" % (k, tuple(A[k].shape), "same bits" if same else "DIFFERS: max abs %g" % float((A[k].float() - Bv[k].float()).abs().max())))
    print("libs [%s] vs [%s]: %s" % (a, b, "identical outputs" if not bad else "%d outputs differ" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
