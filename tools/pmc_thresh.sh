#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in 4 6; do
  OUT=gpurun_out/pmc3_t$t; rm -rf $OUT
  SVO_PERSIST_THRESH=$t rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES --output-format csv -d $OUT -- python3 bench.py --steps 30 --warmup 3 --cpu-seconds 0 --inflight 1 > /dev/null 2>&1
  echo "== thresh $t"; python3 tools/pmc_summary.py $OUT | grep persist_kernel | awk '{print $3, $5}' | tr '\n' ' '; echo
done
