"""GPU box: per-launch table of ONE chain step (crops -> M -> generator -> SoftErosion -> paste-back) outside the generator:
python tools/chain_layers.py [B] -> label, launches, ms, GFLOP, TFLOP/s for every launch whose label is not a generator layer."""
import collections
import csv
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from canonswap_amd import synth
from canonswap_amd.can_swap_e2e import can_swapper
from canonswap_amd.chain import FrameChain

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sds = synth.to_torch(synth.make_state_dicts(0, modules=synth.MODULES + ("motion_extractor",)))
sw = can_swapper(None, state_dicts=sds, max_batch=B)
sw.engine.set_identity(torch.from_numpy(synth.make_identity(7)).cuda(), slot=0)
fc = FrameChain(sw)
crops = torch.from_numpy(np.ascontiguousarray((synth.make_smooth_images(B, seed=2100, size=512).transpose(0, 2, 3, 1) * 255).astype(np.uint8))).cuda()
masks = torch.zeros(B, 512, 512, dtype=torch.uint8, device="cuda"); masks[:, 100:400, 120:380] = 1
ori = torch.randint(0, 256, (B, 1080, 1920, 3), dtype=torch.uint8, device="cuda")
Ms = np.tile(np.array([[0.8, 0, 600.], [0, 0.8, 200.]]), (B, 1, 1))
for _ in range(2):
    fc(crops, masks, Ms, ori, slots=[0] * B)
tmp = tempfile.NamedTemporaryFile(suffix=".csv", delete=False); tmp.close()
os.environ["CANONSWAP_PROFILE_CSV"] = tmp.name
sw.engine.profile_begin()
fc(crops, masks, Ms, ori, slots=[0] * B)
sw.engine.profile_end()
rows = list(csv.DictReader(open(tmp.name)))
agg = collections.OrderedDict()
for r in rows:
    lab = r["label"]
    if not (lab.startswith("M.") or lab.startswith("m_") or lab in ("prepare_crops", "soft_erosion", "paste_back_batch")):
        lab = "(generator)"
    a = agg.setdefault(lab, [0, 0.0, 0.0]); a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["gflop"])
tot = sum(a[1] for a in agg.values())
print(f"one chain step, {B} frames: {tot:.3f} ms of device time")
for k, (n, ms, gf) in agg.items():
    print("%-22s n=%3d %8.3f ms %9.1f GF %8.1f TF/s" % (k, n, ms, gf, gf / ms if ms > 0 else 0))
