#!/bin/bash
# two PMC passes + kernel trace for the library selected by SVO_HIP_LIB; usage: tools/prof_lite.sh <tag> <bench args>
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py "$@" > $OUT/l1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mix -- python3 bench.py "$@" > $OUT/l2.log 2>&1
python3 tools/pmc_summary.py $OUT | grep -E "persist_kernel|fused|stage" | sed 's/(svo::PersistArgs)//' | awk '{ $1=$1; print }' | cut -c1-120
