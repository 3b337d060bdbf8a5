#!/bin/bash
# GPU box: the round's measurement set -> gpurun_out/$1/ (bench line, rocprofv3 kernel stats of the same command, PMC HBM passes)
TAG=${1:-rXX}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /root/repo
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python /root/repo/bench.py --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o pmc --output-format csv -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_MFMA -o pmc --output-format csv -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_MFMA.json 2> $OUT/pmc_MFMA.err
ls -la $OUT $OUT/stats | head -40
du -sh $OUT
