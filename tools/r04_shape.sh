#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_shape.txt; : > $O
for steps in "100 10" "20 5"; do set -- $steps
for i in 3 4 5 6; do for b in 4 5 8; do
  echo -n "steps $1 warmup $2 inflight $i batch $b: " >> $O
  timeout 300 python bench.py --steps $1 --warmup $2 --cpu-seconds 0 --inflight $i --batch $b --isolated 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O 2>&1
done; done; done
cat $O
