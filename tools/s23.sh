#!/bin/bash
# A/B of the shipped library against ab/$1.so: op tests, then alternating bench runs
mkdir -p gpurun_out/s23
python -m pytest tests/test_gpu_ops.py tests/test_gpu_batch32.py -m gpu -x -q 2>&1 | tail -n 3
for i in 1 2 3; do for v in ab/$1.so ""; do
  CANONSWAP_LIB=$v python bench.py --steps 10 --warmup 3 > gpurun_out/s23/b.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/s23/b.json")); print("lib=$v", d["value"], d["roofline"]["frac"], d["ms_per_step"])
PY
done; done
