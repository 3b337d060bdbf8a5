#!/bin/bash
cd /root/repo; O=gpurun_out/s7; mkdir -p $O
run() { echo "== stagger $1 only $2"; CANONSWAP_STAGGER_CYCLES=$1 CANONSWAP_LIB=ab/timeline.so python tools/timeline.py --only $2 --out $O/tl_$2_$1.json 2>&1 | grep -v "amdgpu.ids\|phase cycles"; }
for s in 0 50000 95000 140000; do run $s T.blend; done
for s in 0 25000 48000; do run $s G.c512; done
for s in 0 8000 16000; do run $s G.gb512; done
for s in 0 7000 15000; do run $s v32.c2; done
for s in 0 5000 10000; do run $s v32.c1; done
