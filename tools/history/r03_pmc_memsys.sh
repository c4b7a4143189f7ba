#!/bin/bash
# memory-subsystem counters of the persistent kernel (separate short passes, each under its own timeout)
# usage: tools/r03_pmc_memsys.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; shift
ARGS="--steps 24 --warmup 4 --cpu-seconds 0 --verify 0 --isolated 0 $@"
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
           "TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LEVEL_WAVES SQ_WAVE_CYCLES" \
           "TA_TA_BUSY_sum TD_TD_BUSY_sum TD_TC_STALL_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_RFIFO_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_LFIFO_FULL_sum TCP_CLIENT_UTCL1_INFLIGHT_sum" ; do
  OUT=gpurun_out/pmc_${TAG}_$i; rm -rf $OUT
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $OUT -- python3 bench.py $ARGS > $OUT.log 2>&1
  echo "== $set (rc $?)"; python3 tools/pmc_summary.py $OUT 2>&1 | grep -E "persist_kernel" | awk '{print $3, $4, $5}' | head -12
  grep -m1 "exceeds the capabilities" $OUT.log
  rm -rf $OUT
  i=$((i+1))
done
