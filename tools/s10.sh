#!/bin/bash
cd /root/repo; O=gpurun_out/s10; mkdir -p $O
python -m pytest tests/test_gpu_tail.py tests/test_gpu_bench_ranks.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -40 > $O/pytest_tail.txt; cat $O/pytest_tail.txt
python -m pytest tests/test_gpu_precision.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids > $O/pytest_prec.txt; grep -n "PSNR\|passed\|failed" $O/pytest_prec.txt
