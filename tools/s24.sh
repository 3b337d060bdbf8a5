#!/bin/bash
mkdir -p gpurun_out/s24
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 4
run() { python bench.py $2 > gpurun_out/s24/$3.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/s24/$3.json")); print("$1", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
}
run "b1 default" "--batch 1 --steps 50 --warmup 10" b1
run "b1 latency" "--batch 1 --latency-mode --steps 50 --warmup 10" b1_lat
run "b32" "--steps 10 --warmup 3" b32
run "b1 latency" "--batch 1 --latency-mode --steps 50 --warmup 10" b1_lat
run "b32" "--steps 10 --warmup 3" b32
