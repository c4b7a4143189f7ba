#!/bin/bash
# resident waves split over more launches: (launches in flight) x (waves per CU per launch), four frames per launch
cd $GRAFT_REPO_ROOT
for cfg in "3 10" "4 5" "4 6" "4 8" "5 4" "5 5" "6 4" "6 5" "2 10" "8 3"; do
  set -- $cfg
  echo -n "--inflight $1 --waves $2 -> "
  python bench.py --steps 480 --verify 0 --cpu-seconds 0 --isolated 0 --inflight $1 --waves $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
