#!/bin/bash
# GPU box: the round-6 measurement set -> gpurun_out/$1/: tools/round_profile.sh (bench line, rocprofv3 stats of the same command, PMC passes, families, reconciliation,
# hbm_traffic.json) + the chain's kernel table (rocprofv3 --stats of tools/chain_layers.py) + the one-frame lines + the configs[3]-shape test with its printed rates
TAG=${1:-r06_p}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
bash /root/repo/tools/round_profile.sh $TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/chain_stats -o chain --output-format csv -- python /root/repo/tools/chain_layers.py 64 > $OUT/chain_layers.txt 2> $OUT/chain_stats.err
rm -rf $OUT/chain_stats/*.db $OUT/chain_stats/*kernel_trace.csv
cd /root/repo
python bench.py --batch 1 --latency-mode --steps 200 --warmup 20 --no-fixed-job --no-cpu-baseline > $OUT/bench_b1_latency.json 2> $OUT/b1.err; tail -c 600 $OUT/bench_b1_latency.json
CANONSWAP_PROFILE_CSV=$OUT/layers_b1_latency.csv python bench.py --batch 1 --latency-mode --no-cpu-baseline --no-fixed-job --steps 1 --warmup 3 > /dev/null 2>&1
python tools/layer_table.py $OUT/layers_b1_latency.csv > $OUT/families_b1_latency.txt 2>> $OUT/summarize.err
python -m pytest tests/test_gpu_bench_ranks.py -m gpu -q -s -k configs3 > $OUT/pytest_configs3.txt 2>&1; grep "configs\[3\] shape\|passed\|failed" $OUT/pytest_configs3.txt
python bench.py --frames 1200 --no-cpu-baseline > $OUT/bench_frames1200.json 2>> $OUT/b1.err
python bench.py --force-dist --no-cpu-baseline --no-chain --no-single-frame > $OUT/bench_b64_rccl_world1.json 2>> $OUT/b1.err
du -sh $OUT
