#!/bin/bash
# GPU box: the chain leg -> gpurun_out/$1/: tests of the new entries, the bench line with `chain` / `v2i_body`, rocprofv3 kernel stats of a chain-only run
TAG=${1:-r06_b}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /root/repo
python -m pytest tests/test_gpu_chain.py tests/test_gpu_tail.py tests/test_gpu_motion.py -m gpu -x -q > $OUT/pytest_chain.txt 2>&1; tail -5 $OUT/pytest_chain.txt
python -m pytest tests/test_gpu_latency.py -m gpu -x -q -k knob > $OUT/pytest_knob.txt 2>&1; tail -3 $OUT/pytest_knob.txt
python bench.py --no-single-frame > $OUT/bench.json 2> $OUT/bench.err; tail -c 2500 $OUT/bench.json; tail -5 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fixed-job --no-single-frame > $OUT/bench_profiled.json 2> $OUT/stats.err
rm -rf $OUT/stats/*.db $OUT/stats/*kernel_trace.csv 2>/dev/null
ls $OUT/stats; du -sh $OUT
