#!/bin/bash
# a short form of pmc_mem.sh: TA / TD / TCP busy of the persistent kernel, one frame at a time
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS="--steps 30 --warmup 3 --cpu-seconds 0 --inflight 1 $@"
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TA_BUFFER_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum" "TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  OUT=gpurun_out/pmc_mem2_$i; rm -rf $OUT
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $OUT -- python3 bench.py $ARGS > $OUT.log 2>&1
  python3 tools/pmc_summary.py $OUT 2>&1 | grep -E "persist_kernel" | awk '{print $3, $5}'
  i=$((i+1))
done
