#!/bin/bash
# A/B harness for the GPU box: times tools/gpu_diag.py with each library under ab/ (same box, back to back) and
# optionally collects the LDS bank-conflict counters. Usage: bash tools/ab.sh [pmc] lib1 lib2 ...
PMC=0; if [ "$1" = "pmc" ]; then PMC=1; shift; fi
cp canonswap_amd/libcanonswap_hip.so ab/_work.so
for l in "$@"; do
  cp ab/$l.so canonswap_amd/libcanonswap_hip.so
  for r in 1 2; do
    echo "== $l run $r: $(python tools/gpu_diag.py --batch 16 2>&1 | tail -1)"
  done
  cp gpurun_out/layers_b16.csv gpurun_out/layers_b16_$l.csv
  if [ $PMC = 1 ]; then
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d /root/repo/gpurun_out/pmc_$l -o pmc --output-format csv -- python /root/repo/bench.py --steps 1 --warmup 1 --batch 8 > /root/repo/gpurun_out/pmc_$l.log 2>&1)
  fi
done
cp ab/_work.so canonswap_amd/libcanonswap_hip.so
