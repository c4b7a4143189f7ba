#!/bin/bash
# texture-path counters of the persistent kernel (separate short passes, each under its own timeout:
# a rejected counter set makes rocprofv3 abort and then hang in its finaliser)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS="--steps 30 --warmup 3 --cpu-seconds 0 $@"
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" \
           "TD_TD_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH" ; do
  OUT=gpurun_out/pmc_mem_$i; rm -rf $OUT
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $OUT -- python3 bench.py $ARGS > $OUT.log 2>&1
  echo "== $set (rc $?)"; python3 tools/pmc_summary.py $OUT 2>&1 | grep -E "persist_kernel" | awk '{print $2, $4}' | head -12
  grep -m1 "exceeds the capabilities" $OUT.log
  i=$((i+1))
done
