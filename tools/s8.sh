#!/bin/bash
cd /root/repo; O=gpurun_out/s8; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -q 2>&1 | tail -4
for v in 0 1 0 1; do CANONSWAP_ENC256=$v python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/bench_enc$v.json; python3 -c "
import json; d=json.load(open('$O/bench_enc$v.json')); print('enc256=$v', d['value'], d['roofline']['frac'], d['ms_per_step'])"; done
CANONSWAP_LIB=ab/timeline.so python tools/timeline.py --only enc0 --out $O/timeline.json 2>&1 | grep -v amdgpu.ids
