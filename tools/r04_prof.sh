#!/bin/bash
# round-3 session 2: where the descriptor walk spends its time -- in-kernel section mix (SVO_STAMPS build), VALU /
# wait counters, texture-path and cache counters (separate short PMC passes), refill-threshold x waves sweep
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_prof.txt; : > $O
echo "== stamps (section mix, timeline)" >> $O
SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_stamps.so timeout 300 python tools/r03_timeline.py >> $O 2>&1
SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_stamps.so timeout 300 python tools/stamps.py >> $O 2>&1
ARGS="--steps 40 --warmup 5 --cpu-seconds 0 --verify 0 --isolated 0"
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM" \
           "TA_TA_BUSY_sum TD_TD_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  for d in 1 0; do
    OUT=gpurun_out/pmc_r04_$i; rm -rf $OUT
    SVO_DERIVED=$d timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d $OUT -- python3 bench.py $ARGS > $OUT.log 2>&1
    echo "== derived=$d $set (rc $?)" >> $O; python3 tools/pmc_summary.py $OUT 2>&1 | grep -E "persist_kernel" | head -12 >> $O
    grep -m1 "exceeds the capabilities" $OUT.log >> $O
    rm -rf $OUT
  done
  i=$((i+1))
done
echo "== sweep: waves x threshold (default 4 x 5 in flight)" >> $O
for w in 8 10 12; do for t in 8 9 10 11; do
  echo -n "waves $w thresh $t: " >> $O
  timeout 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --waves $w --thresh $t 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O 2>&1
done; done
for t in 8 9 10 11 12; do
  echo -n "one frame at a time, thresh $t: " >> $O
  timeout 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --inflight 1 --batch 1 --thresh $t 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O 2>&1
done
cat $O
