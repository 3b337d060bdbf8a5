#!/bin/bash
cd /root/repo; O=gpurun_out/s14; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -q -k "spade" 2>&1 | tail -3
CANONSWAP_LIB=ab/cbl2.so python -m pytest tests/test_gpu_ops.py -m gpu -q -k "spade" 2>&1 | tail -3
for lib in "" "ab/cbl2.so"; do for g in 1 2 4 8; do
  CANONSWAP_LIB=$lib CANONSWAP_CBL=$g python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/b.json; python3 -c "
import json; d=json.load(open('$O/b.json')); print('lib=$lib cbl=$g', d['value'], d['roofline']['frac'], d['ms_per_step'])"; done; done
