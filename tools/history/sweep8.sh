#!/bin/bash
# refill threshold in sixteenths (SVO_PERSIST_THRESH overrides svo_set_tuning) x persistent waves per CU, 3 frames in flight
for t in 8 9 10 11; do
  for w in 9 10 11; do
    echo -n "thresh=$t/16 waves/cu=$w: "; SVO_PERSIST_THRESH=$t SVO_PERSIST_WAVES_PER_CU=$w python bench.py --steps 150 --warmup 8 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
