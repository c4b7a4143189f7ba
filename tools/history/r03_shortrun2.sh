#!/bin/bash
cd $GRAFT_REPO_ROOT
for K in "20 5" "200 10"; do
  set -- $K
  for cfg in "20 4 5" "20 5 4" "20 6 4" "20 4 4" "16 4 5" "12 4 5" "10 4 5" "20 3 7" "20 3 8" "16 3 7" "12 3 7" "20 5 5" "20 7 3" "20 10 2" "10 6 4" "10 4 4"; do
    set -- $K $cfg
    echo -n "--steps $1 --warmup $2 --waves $3 --inflight $4 --batch $5 -> "
    python bench.py --cpu-seconds 0 --verify 0 --isolated 0 --steps $1 --warmup $2 --waves $3 --inflight $4 --batch $5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
