#!/bin/bash
# the RCCL gather path at world size 1 (SVO_BENCH_FORCE_COMM=1): ring depth x waves per CU
cd $GRAFT_REPO_ROOT
echo -n "no comm: "; python bench.py --steps 240 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])"
for cfg in "3 10" "4 10" "5 10" "6 10" "4 8" "5 6" "3 9" "3 6"; do
  set -- $cfg
  echo -n "FORCE_COMM --inflight $1 --waves $2 -> "
  SVO_BENCH_FORCE_COMM=1 python bench.py --steps 240 --cpu-seconds 0 --inflight $1 --waves $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])"
done
