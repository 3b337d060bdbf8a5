#!/bin/bash
cd /root/repo; O=gpurun_out/s18; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -q 2>&1 | tail -4
bash tools/round_profile.sh r02_e 2>&1 | tail -3
CANONSWAP_R_SPLIT=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02_e/bench_r_split_off.json
python bench.py --no-cpu-baseline --frames 1200 --steps 2 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r02_e/bench_frames1200.json
CANONSWAP_PROFILE_CSV=/root/repo/gpurun_out/r02_e/layers_b32.csv python bench.py --no-cpu-baseline --steps 1 --warmup 2 > /dev/null 2>&1
CANONSWAP_LIB=ab/timeline.so python tools/timeline.py --out gpurun_out/r02_e/timeline.json 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_e/timeline.txt
