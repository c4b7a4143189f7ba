#!/bin/bash
# throughput vs resident persistent waves per CU, tail amortised over 8 frames per launch (one launch at a time)
cd $GRAFT_REPO_ROOT
for w in 4 6 8 10 12 14 16 18 20; do
  echo -n "--inflight 1 --batch 8 --waves $w -> "
  python bench.py --steps 240 --verify 0 --cpu-seconds 0 --isolated 0 --inflight 1 --batch 8 --waves $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
for w in 3 4 5 6 7 8 10; do
  echo -n "--inflight 3 --batch 4 --waves $w -> "
  python bench.py --steps 240 --verify 0 --cpu-seconds 0 --isolated 0 --inflight 3 --batch 4 --waves $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
