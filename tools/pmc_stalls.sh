#!/bin/bash
# GPU box: wave-state / memory-pipe counters per kernel over a short bench run -> gpurun_out/$1/
TAG=${1:-stalls}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES" \
           "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TA_TCP_STATE_READ TA_BUSY" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc --output-format csv -- python /root/repo/bench.py --steps ${PMC_STEPS:-1} --warmup 1 --no-cpu-baseline --no-fixed-job $BENCH_ARGS > $OUT/p$i.json 2> $OUT/p$i.err
  rm -f $OUT/p$i/pmc_kernel_trace.csv
done
du -sh $OUT
