#!/bin/bash
run() { python bench.py --steps 60 --warmup 6 --cpu-seconds 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for inf in 3; do for w in 6 7 8 9 10 12; do echo -n "inflight=$inf waves/cu=$w: "; SVO_PERSIST_WAVES_PER_CU=$w run --inflight $inf; done; done
for inf in 2; do for w in 12 14; do echo -n "inflight=$inf waves/cu=$w: "; SVO_PERSIST_WAVES_PER_CU=$w run --inflight $inf; done; done
for w in 5 7; do echo -n "inflight=4 waves/cu=$w: "; SVO_PERSIST_WAVES_PER_CU=$w run --inflight 4; done
