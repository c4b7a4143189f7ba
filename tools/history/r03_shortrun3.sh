#!/bin/bash
cd $GRAFT_REPO_ROOT
for K in "10 5" "20 5" "40 5" "60 10" "100 10" "400 10"; do
  for cfg in "10 3 4" "10 4 5" "10 4 6" "10 5 5" "12 4 5"; do
    set -- $K $cfg
    echo -n "--steps $1 --warmup $2 --waves $3 --inflight $4 --batch $5 -> "
    python bench.py --cpu-seconds 0 --verify 0 --isolated 0 --steps $1 --warmup $2 --waves $3 --inflight $4 --batch $5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
for r in "0/8" "0/2"; do for cfg in "10 3 4" "10 4 5"; do for K in "20 5" "400 10"; do
  set -- $K $cfg
  echo -n "--as-rank $r --steps $1 --warmup $2 --waves $3 --inflight $4 --batch $5 -> "
  python bench.py --cpu-seconds 0 --verify 0 --isolated 0 --as-rank $r --steps $1 --warmup $2 --waves $3 --inflight $4 --batch $5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done
