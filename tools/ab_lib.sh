#!/bin/bash
# GPU box: A/B of the shipped library against ab/$1.so (python tools/build_variant.py NAME flags): op tests, then alternating bench runs on the same box
mkdir -p gpurun_out/ab_lib
python -m pytest tests/test_gpu_ops.py tests/test_gpu_batch32.py -m gpu -x -q 2>&1 | tail -n 3
for i in 1 2 3; do for v in ab/$1.so ""; do
  CANONSWAP_LIB=$v python bench.py --steps 10 --warmup 3 > gpurun_out/ab_lib/b.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/ab_lib/b.json")); print("lib=$v", d["value"], d["roofline"]["frac"], d["ms_per_step"])
PY
done; done
