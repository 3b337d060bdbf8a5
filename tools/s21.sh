#!/bin/bash
mkdir -p gpurun_out/s21
for i in 1 2; do for v in 0 1; do
  CANONSWAP_SPADE256=$v python bench.py --steps 10 --warmup 3 > gpurun_out/s21/b_$v.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/s21/b_$v.json")); print("spade256=$v", d["value"], d["roofline"]["frac"], d["ms_per_step"])
PY
done; done
CANONSWAP_SPADE256=1 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -n 2
