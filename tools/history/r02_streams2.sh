#!/bin/bash
cd $GRAFT_REPO_ROOT
export SVO_RING_OWN_STREAMS=1; export GPU_MAX_HW_QUEUES=8
run() { python bench.py --cpu-seconds 0 --verify 0 --isolated 0 --steps 240 --warmup 12 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$*', '->', j['value'], 'Mrays/s', j['ms_per_step'], 'ms')"; }
for i in 3 4 5 6 8; do run --inflight $i; done
for i in 3 4 6; do run --inflight $i --as-rank 0/8 --steps 480; done
for i in 3 4 6; do run --inflight $i --as-rank 0/4 --steps 480; done
run --inflight 4 --waves 8; run --inflight 4 --waves 12; run --inflight 6 --waves 8; run --inflight 6 --waves 6
run --inflight 4 --batch 1; run --inflight 6 --batch 1; run --inflight 8 --batch 1
