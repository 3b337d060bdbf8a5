#!/bin/bash
# GPU box: alternating bench runs of the shipped library under different environment settings (A/B knobs of engine.hip), same box.
#   bash tools/ab_env.sh "" "CANONSWAP_DEPHASE_SPADE=8" "CANONSWAP_DEPHASE_SPADE=16"
mkdir -p gpurun_out/ab_env
for i in 1 2; do for v in "$@"; do
  env $v python bench.py --steps 10 --warmup 3 --no-fixed-job --no-cpu-baseline > gpurun_out/ab_env/b.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/ab_env/b.json")); print("env=[$v]", d["value"], d["roofline"]["frac"], d["ms_per_step"])
PY
done; done
