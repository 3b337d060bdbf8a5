#!/bin/bash
cd /root/repo; O=gpurun_out/s19; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_batch32.py -m gpu -q -x 2>&1 | tail -3
for r in 1 2; do for lib in ab/prev.so ""; do
CANONSWAP_LIB=$lib python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/b.json; python3 -c "
import json; d=json.load(open('$O/b.json')); print('lib=$lib', d['value'], d['roofline']['frac'], d['ms_per_step'])"
done; done
