#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_share.txt; : > $O
for n in 2 4 8; do for b in 5 10 20 40; do
  echo -n "rank 0 of $n, batch $b: " >> $O
  timeout 300 python bench.py --steps 400 --warmup 40 --cpu-seconds 0 --as-rank 0/$n --batch $b --isolated 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])" >> $O 2>&1
done; done
for w in 6 8 12; do
  echo -n "rank 0 of 8, batch 20, waves $w: " >> $O
  timeout 300 python bench.py --steps 400 --warmup 40 --cpu-seconds 0 --as-rank 0/8 --batch 20 --waves $w --isolated 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])" >> $O 2>&1
done
cat $O
