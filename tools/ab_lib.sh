#!/bin/bash
# GPU box: alternating bench runs of two builds of the library on the same box:  bash tools/ab_lib.sh ab/base.so "" [rounds] [tag]
#   ("" = the shipped library); per-layer CSVs of both at the end -> gpurun_out/<tag>/
A=$1; B=$2; R=${3:-2}; TAG=${4:-ab_lib}
cd /root/repo; mkdir -p gpurun_out/$TAG
for i in $(seq $R); do for v in "$A" "$B"; do
  CANONSWAP_LIB=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-fixed-job --no-cpu-baseline --no-chain --no-single-frame > gpurun_out/$TAG/b.json 2>gpurun_out/$TAG/b.err
  python - <<PY | tee -a gpurun_out/$TAG/ab.txt
import json
try:
    d=json.loads(open("gpurun_out/$TAG/b.json").read().strip().splitlines()[-1]); print("lib=[$v]", d["value"], d["roofline"]["frac"], d["ms_per_step"])
except Exception as e: print("lib=[$v] failed", e, open("gpurun_out/$TAG/b.err").read()[-800:])
PY
done; done
k=0; for v in "$A" "$B"; do
  CANONSWAP_LIB=$v CANONSWAP_PROFILE_CSV=/root/repo/gpurun_out/$TAG/layers_$k.csv timeout 600 python bench.py --no-cpu-baseline --no-fixed-job --no-chain --no-single-frame --steps 1 --warmup 2 > /dev/null 2>&1
  k=$((k+1))
done
python tools/cmp_layers.py gpurun_out/$TAG/layers_0.csv gpurun_out/$TAG/layers_1.csv 2>/dev/null | head -16 | tee gpurun_out/$TAG/cmp.txt
