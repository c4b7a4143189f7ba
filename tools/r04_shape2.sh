#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_shape2.txt; : > $O
for rep in 1 2; do
for steps in "20 5" "40 10" "200 10"; do set -- $steps
for ib in "4 5" "5 4" "6 4" "5 8" "6 5" "8 4" "8 3"; do set -- $steps $ib
  echo -n "steps $1 warmup $2 inflight $3 batch $4: " >> $O
  timeout 300 python bench.py --steps $1 --warmup $2 --cpu-seconds 0 --inflight $3 --batch $4 --isolated 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O 2>&1
done; done; done
cat $O
