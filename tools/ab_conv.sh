#!/bin/bash
# op-level A/B on the GPU box: bash tools/ab_conv.sh lib1 lib2 ...   (libraries under ab/, BENCH_ONLY filter honoured)
cp canonswap_amd/libcanonswap_hip.so ab/_work.so
for l in "$@"; do
  cp ab/$l.so canonswap_amd/libcanonswap_hip.so
  echo "== $l"; python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids
done
cp ab/_work.so canonswap_amd/libcanonswap_hip.so
