#!/bin/bash
cd /root/repo; O=gpurun_out/s12; mkdir -p $O
python -m pytest tests/test_gpu_fp8.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids > $O/pytest_fp8.txt; grep -n "frame\|quantisation\|passed\|failed\|Error" $O/pytest_fp8.txt | head -20
for v in "0" "1"; do CANONSWAP_V32_DB=$v python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/bench_v32db$v.json; python3 -c "
import json; d=json.load(open('$O/bench_v32db$v.json')); print('v32db=$v', d['value'], d['roofline']['frac'], d['ms_per_step'])"; done
python bench.py --no-cpu-baseline --steps 10 --fp8-weights --identities 4 2>/dev/null | tail -1 > $O/bench_fp8_id4.json; python3 -c "
import json; d=json.load(open('$O/bench_fp8_id4.json')); print('fp8 id4', d['value'], d['roofline']['frac'], d['dtype'], d['config']['identities_resident'])"
