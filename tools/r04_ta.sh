#!/bin/bash
# texture-path / memory counters of the persistent kernel, both walks, two or three counters per pass (more are refused)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ta.txt; : > $O
ARGS="--steps 40 --warmup 4 --cpu-seconds 0 --verify 0 --isolated 0"
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  for d in 1 0; do
    OUT=gpurun_out/pmc_ta_$i; rm -rf $OUT
    SVO_DERIVED=$d timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d $OUT -- python3 bench.py $ARGS > $OUT.log 2>&1
    echo "== derived=$d $set (rc $?)" >> $O; python3 tools/pmc_summary.py $OUT 2>&1 | grep -E "persist_kernel" | head -8 >> $O
    grep -m1 "exceeds the capabilities" $OUT.log >> $O
    rm -rf $OUT $OUT.log
  done
  i=$((i+1))
done
cat $O
