#!/bin/bash
# GPU session 1: sanity (tests), XCD-map A/B on the whole step, per-wave timelines
cd /root/repo; O=gpurun_out/s1; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
for m in 0 1 2 0 1 2; do
  CANONSWAP_XCD_MAP=$m python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/bench_xcd${m}_$RANDOM.json
done
grep -h -o '"value": [0-9.]*\|"frac": [0-9.]*' $O/bench_xcd*.json | paste - - - -
for m in 0 1 2; do
  CANONSWAP_XCD_MAP=$m CANONSWAP_LIB=ab/timeline.so python tools/timeline.py --out $O/timeline_xcd$m.json 2>&1 | grep -v amdgpu.ids | tee $O/timeline_xcd$m.txt
done
