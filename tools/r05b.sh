#!/bin/bash
# round 5, call b: pricing micro-benchmark, batch sweep past 64, PSNR survey of the whole 256-frame pool
O=gpurun_out/r05b; mkdir -p $O
tools/bin/mfma_power3 1.0 > $O/mfma_power3.txt 2>&1
for b in 64 72 80 84; do python bench.py --batch $b --no-cpu-baseline --no-fixed-job --steps 16 --warmup 3 2>/dev/null | tail -1 > $O/bench_b$b.json; done
python bench.py --batch 80 --steps 4 --warmup 1 2>$O/b80.err | tail -1 > $O/bench_b80_parity.json
python tests/diag/psnr_pool.py 64 $O/psnr_worst_frame.json 4 > $O/psnr_pool256.txt 2>&1
tail -2 $O/psnr_pool256.txt; cat $O/mfma_power3.txt
for b in 64 72 80 84; do python -c "
import json,sys
d=json.load(open('$O/bench_b$b.json')); print($b, d['value'], d['roofline']['frac'], d['roofline']['end_to_end_frac'])"; done
python -c "
import json
d=json.load(open('$O/bench_b80_parity.json')); print(d['value'], d.get('psnr_db_min'), d.get('parity_sample'))"
