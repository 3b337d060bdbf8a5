cd /root/repo; mkdir -p gpurun_out/r04z
for v in "X=1" "CANONSWAP_SHORTCUT_ALGEBRA=0" "CANONSWAP_SHORTCUT_FUSE=0" "CANONSWAP_WIDE=0" "X=2"; do
 for m in "" "--latency-mode"; do
  env $v python bench.py --batch 1 --steps 60 --warmup 8 --no-cpu-baseline --no-fixed-job $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$m', d['ms_per_step'])" | tee -a gpurun_out/r04z/lat_knobs.txt
 done
done
