#!/bin/bash
run() { python bench.py --steps 100 --warmup 6 --cpu-seconds 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for t in 4 5 6 7; do for w in 10 12 14; do echo -n "thresh=$t waves/cu=$w: "; SVO_PERSIST_THRESH=$t SVO_PERSIST_WAVES_PER_CU=$w run; done; done
for t in 4 5 6; do echo -n "inflight=1 thresh=$t: "; SVO_PERSIST_THRESH=$t run --inflight 1; done
