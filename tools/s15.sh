#!/bin/bash
cd /root/repo; O=gpurun_out/s15; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for r in 1 2; do
CANONSWAP_LIB=ab/nocbl.so CANONSWAP_CBL=1 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/b.json; python3 -c "
import json; d=json.load(open('$O/b.json')); print('nocbl', d['value'], d['roofline']['frac'], d['ms_per_step'])"
python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/b.json; python3 -c "
import json; d=json.load(open('$O/b.json')); print('cbl8+hilo', d['value'], d['roofline']['frac'], d['ms_per_step'])"
done
CANONSWAP_PROFILE_CSV=$O/layers_b32.csv python bench.py --no-cpu-baseline --steps 1 --warmup 2 > /dev/null 2>&1
