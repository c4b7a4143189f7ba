#!/bin/bash
# the driver's invocation is short (--steps 20 --warmup 5): which (waves per CU, launches in flight, frames per launch)
# loses least to the start and the drain of the timed region?
cd $GRAFT_REPO_ROOT
for K in "20 5" "200 10"; do
  set -- $K
  for cfg in "10 3 4" "20 3 4" "20 2 4" "16 3 4" "20 2 5" "20 2 10" "10 2 10" "20 1 20" "14 3 4" "20 3 7" "20 4 5" "20 3 2" "20 3 1" "10 3 1"; do
    set -- $K $cfg
    echo -n "--steps $1 --warmup $2 --waves $3 --inflight $4 --batch $5 -> "
    python bench.py --cpu-seconds 0 --verify 0 --isolated 0 --steps $1 --warmup $2 --waves $3 --inflight $4 --batch $5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
