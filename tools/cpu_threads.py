"""The oracle (fp32 PyTorch-CPU restatement of the path) on the GPU box's host cores at several thread counts: frames/s of one-frame forwards.
The record behind bench.py's cpu_baseline using 32 threads.   python tools/cpu_threads.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonswap_amd import synth  # noqa: E402
from oracle import canonswap_ref as O  # noqa: E402

sds = synth.to_torch(synth.make_state_dicts(0))
inp = synth.make_frame_inputs(1, seed=1000, size=256)
a = [torch.from_numpy(inp[k]) for k in ("img", "x_t", "x_can")]
idv = torch.from_numpy(synth.make_identity(7))
print(f"host cores {os.cpu_count()}")
for th in (8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1):
        break
    torch.set_num_threads(th)
    with torch.no_grad():
        O.swap_frame(sds, *a, idv)
        t = time.perf_counter()
        n = 3
        for _ in range(n):
            O.swap_frame(sds, *a, idv)
        dt = time.perf_counter() - t
    print(f"threads {th:4d}: {n / dt:6.3f} frames/s ({dt / n:5.2f} s per frame)", flush=True)
