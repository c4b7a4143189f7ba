#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_waves1.txt; : > $O
for w in 8 10 12 14 16 18 20 24 0; do
  echo -n "one frame at a time, waves $w: " >> $O
  timeout 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --inflight 1 --batch 1 --waves $w --isolated 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O 2>&1
done
cat $O
