#!/bin/bash
# GPU box: one-frame latency-mode A/B of library builds:  bash tools/ab_lat.sh TAG "" ab/x.so ab/y.so ...   ("" = the shipped library)
# per build: ms per frame (bench.py --batch 1 --latency-mode), then the per-family table of one profiled step -> gpurun_out/TAG/
TAG=$1; shift
cd /root/repo; O=gpurun_out/$TAG; mkdir -p $O
for r in 1 2; do for v in "$@"; do
  n=$(basename "${v:-base}" .so)
  CANONSWAP_LIB=$v timeout 300 python bench.py --batch 1 --latency-mode --steps 60 --warmup 10 --no-cpu-baseline --no-fixed-job --no-chain --no-single-frame 2>$O/$n.err | tail -1 > $O/$n.json
  python - <<PY | tee -a $O/ab.txt
import json
try:
    d=json.loads(open("$O/$n.json").read().strip().splitlines()[-1]); print("$n", d["ms_per_step"], d["roofline"]["frac"])
except Exception as e: print("$n failed", e, open("$O/$n.err").read()[-600:])
PY
done; done
for v in "$@"; do
  n=$(basename "${v:-base}" .so)
  CANONSWAP_LIB=$v CANONSWAP_PROFILE_CSV=/root/repo/$O/layers_$n.csv timeout 300 python bench.py --batch 1 --latency-mode --no-cpu-baseline --no-fixed-job --no-chain --no-single-frame --steps 1 --warmup 2 > /dev/null 2>&1
  python tools/layer_table.py $O/layers_$n.csv > $O/families_$n.txt
done
