#!/bin/bash
# one frame at a time: throughput vs resident waves per CU (what a 24-wave kernel could be worth)
for w in 8 12 16 18 20; do
  echo -n "inflight=1 waves/cu=$w: "; SVO_PERSIST_WAVES_PER_CU=$w SVO_PERSIST_THRESH=9 python bench.py --steps 80 --warmup 6 --cpu-seconds 0 --inflight 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
