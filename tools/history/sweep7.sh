#!/bin/bash
for w in 8 9 10 11 12 14; do
  echo -n "inflight=3 waves/cu=$w: "; SVO_PERSIST_WAVES_PER_CU=$w SVO_PERSIST_THRESH=9 python bench.py --steps 150 --warmup 8 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
