#!/bin/bash
# quick knob sweep of the persistent pipeline on the GPU box
for t in 3 4 5 6; do
  echo -n "thresh=$t: "; SVO_PERSIST_THRESH=$t python bench.py --steps 20 --warmup 3 --pipeline 1 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
for w in 12 16; do
  echo -n "waves/cu=$w: "; SVO_PERSIST_WAVES_PER_CU=$w python bench.py --steps 20 --warmup 3 --pipeline 1 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
