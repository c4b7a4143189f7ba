#!/bin/bash
# tuning sweep with distinct frames per step (bench.py --waves / --thresh / --inflight), and the strong-scaling what-ifs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_sweep; mkdir -p $O
run() { python bench.py --cpu-seconds 0 --verify 0 --steps 300 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$*', '->', j['value'], 'Mrays/s', j['ms_per_step'], 'ms')"; }
for w in 8 10 12 14; do for t in 8 9 10; do run --waves $w --thresh $t --inflight 3; done; done
for i in 2 4 5; do run --waves 10 --thresh 9 --inflight $i; done
run --waves 8 --thresh 9 --inflight 4
run --waves 6 --thresh 9 --inflight 5
echo "--- what-if: rank 0 of 8"
for w in 3 4 6 8; do for i in 3 4 6; do run --as-rank 0/8 --waves $w --inflight $i; done; done
echo "--- what-if: rank 0 of 4"
for w in 4 6 8; do for i in 3 4; do run --as-rank 0/4 --waves $w --inflight $i; done; done
echo "--- what-if: rank 0 of 2"
for w in 6 8 10; do run --as-rank 0/2 --waves $w --inflight 3; done
