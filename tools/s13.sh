#!/bin/bash
cd /root/repo; O=gpurun_out/s13; mkdir -p $O
python -m pytest tests/test_gpu_latency.py tests/test_gpu_batch32.py tests/test_gpu_stages.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -12
for v in "" "--latency-mode"; do python bench.py --no-cpu-baseline --batch 1 --steps 50 --warmup 10 $v 2>/dev/null | tail -1 > $O/bench_b1$v.json; python3 -c "
import json; d=json.load(open('$O/bench_b1$v.json')); print('b1 $v', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
python - <<'PY'
import os, sys
sys.path.insert(0, '/root/repo')
import torch
from canonswap_amd import synth
from canonswap_amd.can_swap_e2e import can_swapper
sds = synth.to_torch(synth.make_state_dicts(0))
sw = can_swapper(None, state_dicts=sds, max_batch=1, latency_mode=True)
inp = synth.make_frame_inputs(1, seed=3000, size=256)
ga = [torch.from_numpy(inp[k]).cuda() for k in ("img", "x_t", "x_can")]
idv = torch.from_numpy(synth.make_identity(7)).cuda()
for _ in range(3): sw.engine.swap_frames(*ga, idv)
os.environ["CANONSWAP_PROFILE_CSV"] = "/root/repo/gpurun_out/s13/layers_b1_latency.csv"
sw.engine.profile_begin(); sw.engine.swap_frames(*ga, idv); p = sw.engine.profile_end()
print("profiled: conv %.2f ms (%d launches) other %.2f ms (%d launches)" % (p["conv_ms"], p["conv_launches"], p["other_ms"], p["other_launches"]))
PY
