#!/bin/bash
mkdir -p gpurun_out/s22
python -m pytest tests/test_gpu_ops.py tests/test_gpu_batch32.py -m gpu -x -q 2>&1 | tail -n 5
for i in 1 2; do for v in ab/nopair.so ""; do
  CANONSWAP_LIB=$v python bench.py --steps 10 --warmup 3 > gpurun_out/s22/b.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/s22/b.json")); print("lib=$v", d["value"], d["roofline"]["frac"], d["ms_per_step"])
PY
done; done
python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5
