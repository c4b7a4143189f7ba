#!/bin/bash
# frames in flight x waves per CU (threshold 5/8)
for f in 2 3 4; do
  for w in 7 10 13; do
    echo -n "inflight=$f waves/cu=$w: "; SVO_PERSIST_WAVES_PER_CU=$w SVO_PERSIST_THRESH=5 python bench.py --steps 120 --warmup 8 --cpu-seconds 0 --inflight $f 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
