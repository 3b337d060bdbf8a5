#!/bin/bash
cd /root/repo; O=gpurun_out/s11; mkdir -p $O
python tools/diag_precision.py 3 2>&1 | grep -v amdgpu.ids | tee $O/diag.txt
python -m pytest tests/test_gpu_precision.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids > $O/pytest_prec.txt; grep -n "PSNR\|passed\|failed" $O/pytest_prec.txt
for v in 0 1 0 1; do CANONSWAP_R_SPLIT=$v python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/bench_rs$v.json; python3 -c "
import json; d=json.load(open('$O/bench_rs$v.json')); print('r_split=$v', d['value'], d['roofline']['frac'], d['ms_per_step'])"; done
python -m pytest tests -m gpu -q 2>&1 | tail -5
