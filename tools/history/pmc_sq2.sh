#!/bin/bash
# SQ-side counters of the persistent kernel at the bench tuning (waves/CU 10, threshold 5/8), one frame at a time
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export SVO_PERSIST_THRESH=9 SVO_PERSIST_WAVES_PER_CU=${WAVES:-10}
ARGS="--steps 30 --warmup 3 --cpu-seconds 0 --inflight 1"
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_IFETCH SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU" ; do
  OUT=gpurun_out/pmc_sq2_$i; rm -rf $OUT
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $OUT -- python3 bench.py $ARGS > $OUT.log 2>&1
  echo "== pass $i (rc $?)"; python3 tools/pmc_summary.py $OUT 2>&1 | grep -E "persist_kernel" | awk '{print $3, $5}'
  grep -m1 "exceeds the capabilities" $OUT.log
  tail -1 $OUT.log | cut -c1-120
  i=$((i+1))
done
