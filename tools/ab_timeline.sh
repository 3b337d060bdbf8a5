#!/bin/bash
# epilogue A/B under the per-wave timeline: bash tools/ab_timeline.sh variant...   (libraries ab/<variant>.so built with -DCS_TIMELINE)
mkdir -p gpurun_out/ab_timeline
for v in "$@"; do
  echo "== $v"
  CANONSWAP_LIB=ab/$v.so timeout 300 python tools/timeline.py --out gpurun_out/ab_timeline/$v.json 2>&1 | grep -v "^phase\|amdgpu.ids"
done
