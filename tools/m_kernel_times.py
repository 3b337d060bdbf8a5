"""GPU box: per-launch times of M's own kernels over one chain step (hipEvent profile of the engine), by label and occurrence:
python tools/m_kernel_times.py [B]"""
import csv
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from canonswap_amd import synth
from canonswap_amd.can_swap_e2e import can_swapper

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sds = synth.to_torch(synth.make_state_dicts(0, modules=synth.MODULES + ("motion_extractor",)))
sw = can_swapper(None, state_dicts=sds, max_batch=B)
img = torch.from_numpy(synth.make_smooth_images(B, seed=77, size=256)).cuda()
for _ in range(3):
    sw.engine.motion_extract_raw(img)
tmp = tempfile.NamedTemporaryFile(suffix=".csv", delete=False); tmp.close()
os.environ["CANONSWAP_PROFILE_CSV"] = tmp.name
sw.engine.profile_begin()
sw.engine.motion_extract_raw(img)
sw.engine.profile_end()
rows = list(csv.DictReader(open(tmp.name)))
seen = {}
for r in rows:
    if r["label"].startswith("m_"):
        seen.setdefault(r["label"], []).append(float(r["ms"]) * 1e3)
for k, v in seen.items():
    print("%-10s n=%2d total %.3f ms  per launch (us): %s" % (k, len(v), sum(v) / 1e3, " ".join("%.0f" % x for x in v)))
print("M total %.3f ms" % sum(float(r["ms"]) for r in rows))
