#!/bin/bash
# GPU box: the round's measurement set -> gpurun_out/$1/ (bench line, rocprofv3 kernel stats of the same command, PMC HBM / MFMA passes,
# per-kernel PMC summary, hbm_traffic.json for bench.py's roofline.traffic, rocprof <-> HIP-event reconciliation)
TAG=${1:-rXX}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /root/repo
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --no-fixed-job --no-chain --no-single-frame > $OUT/bench_profiled.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o pmc --output-format csv -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fixed-job --no-chain --no-single-frame > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_MFMA -o pmc --output-format csv -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fixed-job --no-chain --no-single-frame > $OUT/pmc_MFMA.json 2> $OUT/pmc_MFMA.err
cd /root/repo
CANONSWAP_PROFILE_CSV=$OUT/layers_b64.csv python bench.py --no-cpu-baseline --no-fixed-job --no-chain --no-single-frame --steps 1 --warmup 2 > /dev/null 2>&1
python tools/layer_table.py $OUT/layers_b64.csv > $OUT/families.txt 2>> $OUT/summarize.err
python tools/summarize_pmc.py $OUT > $OUT/pmc_summary.csv 2> $OUT/summarize.err
BATCH=$(python -c "import json,sys; print(json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])['config']['frames_per_launch_per_gpu'])")
python tools/summarize_pmc.py $OUT $BATCH $OUT/hbm_traffic.json 2>> $OUT/summarize.err
python tools/reconcile_profile.py $(ls $OUT/stats/*kernel_stats.csv | head -1) $OUT/bench_profiled.json 43 > $OUT/reconcile.txt 2>> $OUT/summarize.err
cat $OUT/reconcile.txt | head -8; cat $OUT/hbm_traffic.json
# keep the merge-back under the gpurun limit: the raw counter dumps are large
rm -rf $OUT/pmc_FETCH_SIZE/*.db $OUT/pmc_WRITE_SIZE/*.db $OUT/pmc_MFMA/*.db $OUT/stats/*.db 2>/dev/null
# ... and the per-dispatch dumps the summaries above were made from (tens of MB since the default line carries the chain and one-frame legs)
rm -f $OUT/pmc_*/*counter_collection.csv $OUT/pmc_*/*kernel_trace.csv $OUT/stats/*kernel_trace.csv 2>/dev/null
du -sh $OUT
