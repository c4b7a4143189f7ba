#!/bin/bash
# use_beam = 1: does leaving CU slots free for the coarse pass of the next dispatch pay?  (waves per CU per launch x launches in flight)
cd $GRAFT_REPO_ROOT
for b in 4 8; do for w in 7 8 9 10; do for f in 3 4; do
  echo -n "--beam 1 --batch $b --waves $w --inflight $f -> "
  python bench.py --steps 480 --verify 0 --cpu-seconds 0 --isolated 0 --beam 1 --batch $b --waves $w --inflight $f 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done
