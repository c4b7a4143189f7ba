#!/bin/bash
# profile bench.py on the GPU box: kernel trace + three PMC passes (separate runs, as the guide prescribes)
# (each pass under its own timeout: a rejected counter set makes rocprofv3 abort and hang in its finaliser)
# usage: tools/prof.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py "$@" > $OUT/bench_trace.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py "$@" > $OUT/bench_pmc_sq.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/pmc_fetch -- python3 bench.py "$@" > $OUT/bench_pmc_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_write -- python3 bench.py "$@" > $OUT/bench_pmc_write.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mix -- python3 bench.py "$@" > $OUT/bench_pmc_mix.log 2>&1
find $OUT -name "*.csv" | head -30
tail -2 $OUT/bench_trace.log | cut -c1-300
