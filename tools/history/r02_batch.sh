#!/bin/bash
# frame batching: parity, then frames per launch x launches in flight x waves per CU, and the strong-scaling what-ifs with it
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_inflight.py -x -q -m gpu 2>&1 | tail -3
run() { python bench.py --cpu-seconds 0 --steps 240 --warmup 12 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$*', '->', j['value'], 'Mrays/s', j['ms_per_step'], 'ms', j['verified'])"; }
for B in 1 2 4 6; do for i in 2 3; do run --batch $B --inflight $i; done; done
run --batch 4 --inflight 3 --waves 12
run --batch 4 --inflight 3 --waves 14
run --batch 4 --inflight 2 --waves 16
run --batch 8 --inflight 2 --waves 20
run --batch 4 --inflight 1 --waves 0
run --batch 8 --inflight 1 --waves 0
echo "--- what-if rank 0 of N with batching"
for n in 8 4 2; do for B in 1 2 4 8; do run --as-rank 0/$n --batch $B; done; done
