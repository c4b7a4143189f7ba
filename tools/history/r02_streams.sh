#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --verify 0 --isolated 0 --steps 240 --warmup 12 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$SVO_RING_OWN_STREAMS $GPU_MAX_HW_QUEUES $*', '->', j['value'], 'Mrays/s', j['ms_per_step'], 'ms')"; }
for own in 0 1; do export SVO_RING_OWN_STREAMS=$own; for i in 3 4 5 6; do run --inflight $i; done; done
export SVO_RING_OWN_STREAMS=1; export GPU_MAX_HW_QUEUES=8; for i in 3 4 6; do run --inflight $i; done
