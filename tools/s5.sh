#!/bin/bash
cd /root/repo; O=gpurun_out/$1; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
for m in 0 2; do CANONSWAP_XCD_MAP=$m python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/bench_xcd$m.json; python3 -c "
import json; d=json.load(open('$O/bench_xcd$m.json')); print('xcd$m', d['value'], d['roofline']['frac'], d['ms_per_step'], d['roofline']['other_kernels_ms_per_step'])"; done
CANONSWAP_PROFILE_CSV=$O/layers_b32.csv python bench.py --no-cpu-baseline --steps 1 --warmup 2 > /dev/null 2>&1
CANONSWAP_LIB=ab/timeline.so python tools/timeline.py --out $O/timeline.json 2>&1 | grep -v amdgpu.ids | tee $O/timeline.txt
