#!/bin/bash
cd /root/repo; O=gpurun_out/s2; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
CANONSWAP_LIB=ab/timeline.so python tools/timeline.py --out $O/timeline.json 2>&1 | grep -v amdgpu.ids | tee $O/timeline.txt
CANONSWAP_XCD_MAP=2 CANONSWAP_LIB=ab/timeline.so python tools/timeline.py --out $O/timeline_xcd2.json 2>&1 | grep -v amdgpu.ids | tee $O/timeline_xcd2.txt
