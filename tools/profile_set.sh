#!/bin/bash
# GPU box: smoke, the GPU test suite and the full measurement set of a round -> gpurun_out/$TAG/   (bash tools/profile_set.sh r02_g; needs ab/timeline.so = python tools/build_variant.py timeline -DCS_TIMELINE)
TAG=${1:-rXX}
cd /root/repo; O=gpurun_out/profile_set; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -q 2>&1 | tail -4
bash tools/round_profile.sh ${TAG} 2>&1 | tail -3
python bench.py --no-cpu-baseline --frames 1200 --steps 2 --warmup 1 2>/dev/null | tail -1 > gpurun_out/${TAG}/bench_frames1200.json
CANONSWAP_PROFILE_CSV=/root/repo/gpurun_out/${TAG}/layers_b32.csv python bench.py --no-cpu-baseline --steps 1 --warmup 2 > /dev/null 2>&1
CANONSWAP_LIB=ab/timeline.so python tools/timeline.py --out gpurun_out/${TAG}/timeline.json 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}/timeline.txt
python bench.py --no-cpu-baseline --batch 1 --steps 50 --warmup 10 2>/dev/null | tail -1 > gpurun_out/${TAG}/bench_b1.json
python bench.py --no-cpu-baseline --batch 1 --latency-mode --steps 50 --warmup 10 2>/dev/null | tail -1 > gpurun_out/${TAG}/bench_b1_latency_mode.json
CANONSWAP_PROFILE_CSV=/root/repo/gpurun_out/${TAG}/layers_b1_latency_mode.csv python bench.py --no-cpu-baseline --batch 1 --latency-mode --steps 1 --warmup 2 > /dev/null 2>&1
python bench.py --no-cpu-baseline --identities 4 2>/dev/null | tail -1 > gpurun_out/${TAG}/bench_4identities.json
