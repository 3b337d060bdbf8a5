#!/bin/bash
# epilogue A/B under the per-wave timeline: bash tools/s20.sh variant...   (libraries ab/<variant>.so built with -DCS_TIMELINE)
mkdir -p gpurun_out/s20
for v in "$@"; do
  echo "== $v"
  CANONSWAP_LIB=ab/$v.so timeout 300 python tools/timeline.py --out gpurun_out/s20/$v.json 2>&1 | grep -v "^phase\|amdgpu.ids"
done
