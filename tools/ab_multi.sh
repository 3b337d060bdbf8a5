#!/bin/bash
# GPU box: alternating bench runs of the shipped library and ab/<name>.so for every name given, no tests
mkdir -p gpurun_out/ab_lib
for i in 1 2 3; do for n in "" "$@"; do
  v=""; [ -n "$n" ] && v=ab/$n.so
  CANONSWAP_LIB=$v python bench.py --steps 10 --warmup 3 --no-fixed-job --no-cpu-baseline > gpurun_out/ab_lib/b.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/ab_lib/b.json")); print("lib=$v", d["value"], d["roofline"]["frac"], d["ms_per_step"])
PY
done; done
