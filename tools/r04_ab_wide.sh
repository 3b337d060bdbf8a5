#!/bin/bash
# GPU box: parity of conv_wide, then alternating bench runs with / without it on the same box, then per-layer CSVs of both
TAG=${1:-r04a}
cd /root/repo; mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -5 | tee gpurun_out/$TAG/pytest_wide.txt
for i in 1 2; do for v in "CANONSWAP_WIDE=0" "CANONSWAP_WIDE=1"; do
  env $v timeout 600 python bench.py --steps 10 --warmup 3 --no-fixed-job --no-cpu-baseline > gpurun_out/$TAG/b.json 2>gpurun_out/$TAG/b.err
  python - <<PY | tee -a gpurun_out/$TAG/ab.txt
import json
try:
    d=json.loads(open("gpurun_out/$TAG/b.json").read().strip().splitlines()[-1]); print("env=[$v]", d["value"], d["roofline"]["frac"], d["ms_per_step"], d.get("psnr_db_min"))
except Exception as e: print("env=[$v] failed", e, open("gpurun_out/$TAG/b.err").read()[-1500:])
PY
done; done
for v in 0 1; do
  CANONSWAP_WIDE=$v CANONSWAP_PROFILE_CSV=/root/repo/gpurun_out/$TAG/layers_wide$v.csv timeout 600 python bench.py --no-cpu-baseline --no-fixed-job --steps 1 --warmup 2 > /dev/null 2>&1
  python tools/layer_table.py gpurun_out/$TAG/layers_wide$v.csv | head -12 | tee gpurun_out/$TAG/table_wide$v.txt
done
python tools/cmp_layers.py gpurun_out/$TAG/layers_wide0.csv gpurun_out/$TAG/layers_wide1.csv 2>/dev/null | head -70 > gpurun_out/$TAG/cmp.txt
