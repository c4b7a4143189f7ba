#!/bin/bash
# re-tune with 4 frames per launch
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --verify 0 --isolated 0 --steps 240 --warmup 12 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$*', '->', j['value'], 'Mrays/s', j['ms_per_step'], 'ms')"; }
for t in 8 9 10 11; do for w in 8 10; do run --thresh $t --waves $w; done; done
run --batch 3; run --batch 5; run --batch 8
run --batch 4 --inflight 4
run --batch 2 --inflight 4
