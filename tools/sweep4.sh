#!/bin/bash
# knob sweep with frames in flight (bench defaults): SVO_PERSIST_WAVES_PER_CU x SVO_PERSIST_THRESH
for w in 8 10 12 16 20; do
  for t in 4 5 6; do
    echo -n "waves/cu=$w thresh=$t: "; SVO_PERSIST_WAVES_PER_CU=$w SVO_PERSIST_THRESH=$t python bench.py --steps 100 --warmup 6 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
