#!/bin/bash
# One GPU-box session: named steps run in order, each logged to gpurun_out/<tag>_<step kind>.txt.
#   usage (from the repository root, through gpurun):  bash tools/gpu_session.sh <tag> <step> [<step> ...]
#   steps:  tests:<pytest args>      python -m pytest <args> -q -m gpu
#           ab:<variants,comma>      interleaved A/B of libsvohip_<variant>.so builds against the default library
#                                    (ROUNDS passes, BENCH_ARGS; tools/ab_interleaved.sh)
#           argsab:<args|args|...>   interleaved A/B of bench.py argument sets (tools/args_ab.sh)
#           envab:<VAR=a,VAR=b,->    interleaved A/B of environment settings on the default library (tools/env_ab.sh)
#           bench:<bench args>       one bench line
#           prof:<bench args>        rocprofv3 --kernel-trace --stats of the bench command
#           pmc:<bench args>         tools/pmc_pass.py (separate counter passes)
#           sh:<command>             anything else
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=$1; shift
mkdir -p gpurun_out
for step in "$@"; do
  kind=${step%%:*}; arg=${step#*:}
  out=gpurun_out/${TAG}_${kind}.txt
  echo "##### $step" >> $out
  case $kind in
    tests) timeout 3000 python -m pytest $arg -q -m gpu -x 2>&1 | tail -15 >> $out ;;
    ab) bash tools/ab_interleaved.sh ${arg//,/ } >> $out 2>&1 ;;
    argsab) IFS='|' read -ra SETS <<< "$arg"; bash tools/args_ab.sh "${SETS[@]}" >> $out 2>&1 ;;
    envab) bash tools/env_ab.sh ${arg//,/ } >> $out 2>&1 ;;
    bench) timeout 900 python bench.py $arg 2>&1 | tail -1 >> $out ;;
    prof) (cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o trace -- python3 $GRAFT_REPO_ROOT/bench.py $arg > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1); find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" -exec head -12 {} \; >> $out ;;
    pmc) timeout 1800 python tools/pmc_pass.py --tag $TAG -- $arg >> $out 2>&1 ;;
    sh) timeout 3000 bash -c "$arg" >> $out 2>&1 ;;
    *) echo "unknown step $step" >> $out ;;
  esac
done
tail -n 40 gpurun_out/${TAG}_*.txt
