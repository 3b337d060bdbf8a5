#!/bin/bash
cd /root/repo; O=gpurun_out/s17; mkdir -p $O
for fam in base heavy; do
python - <<PY
import os, sys
sys.path.insert(0, '/root/repo')
import torch
from canonswap_amd import synth
from canonswap_amd.can_swap_e2e import can_swapper
fam = "$fam"
sds = synth.to_torch(synth.make_state_dicts(0, family="heavy_tail" if fam == "heavy" else "uniform"))
sw = can_swapper(None, state_dicts=sds, max_batch=2)
inp = synth.make_frame_inputs(2, seed=1000, size=256)
ga = [torch.from_numpy(inp[k]).cuda() for k in ("img", "x_t", "x_can")]
idv = torch.from_numpy(synth.make_identity(7)).cuda()
sw.engine.swap_frames(*ga, idv)
os.environ["CANONSWAP_PROFILE_CSV"] = "/root/repo/gpurun_out/s17/amax_%s.csv" % fam
sw.engine.profile_begin(); sw.engine.swap_frames(*ga, idv); sw.engine.profile_end()
PY
done
python3 - <<'PY'
import csv
for fam in ("base", "heavy"):
    rows = [r for r in csv.DictReader(open("/root/repo/gpurun_out/s17/amax_%s.csv" % fam)) if r.get("fp16_amax")]
    rows.sort(key=lambda r: -float(r["fp16_amax"]))
    print(fam, len(rows), "layers with fp16 outputs; largest:", [(r["label"], float(r["fp16_amax"])) for r in rows[:6]])
PY
