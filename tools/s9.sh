#!/bin/bash
cd /root/repo; O=gpurun_out/s9; mkdir -p $O
python -m pytest tests/test_gpu_precision.py tests/test_gpu_bench_ranks.py tests/test_gpu_identity.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids | tail -30 > $O/pytest.txt; cat $O/pytest.txt
python bench.py --steps 10 2>/dev/null | tail -1 > $O/bench.json; cat $O/bench.json
python bench.py --steps 3 --frames 300 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f300.json; cat $O/bench_f300.json
