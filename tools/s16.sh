#!/bin/bash
cd /root/repo; O=gpurun_out/s16; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for r in 1 2; do for lib in ab/prev.so ""; do
CANONSWAP_LIB=$lib python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/b.json; python3 -c "
import json; d=json.load(open('$O/b.json')); print('lib=$lib', d['value'], d['roofline']['frac'], d['ms_per_step'])"
done; done
CANONSWAP_PROFILE_CSV=$O/layers_b32.csv python bench.py --no-cpu-baseline --steps 1 --warmup 2 > /dev/null 2>&1
