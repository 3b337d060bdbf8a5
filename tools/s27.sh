#!/bin/bash
mkdir -p gpurun_out/s27
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_batch32.py -m gpu -x -q 2>&1 | tail -n 3
ls tests | grep gpu | head -30
CANONSWAP_PROFILE_CSV=/root/repo/gpurun_out/s27/layers.csv python bench.py --no-cpu-baseline --steps 1 --warmup 2 > /dev/null 2>&1
grep -v "^0," gpurun_out/s27/layers.csv | awk -F, '{a[$2]+=$3; n[$2]++} END{for(k in a) printf "%-20s %3d %8.3f\n", k, n[k], a[k]}' | sort -k3 -n -r
for i in 1 2; do python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['other_kernels_ms_per_step'])"; done
