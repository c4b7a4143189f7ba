#!/bin/bash
# waves per CU x refill threshold under the 4 x 5 default
cd $GRAFT_REPO_ROOT
for w in 9 10 11 12; do for t in 8 9 10; do
  echo -n "--waves $w --thresh $t -> "
  for K in "200 10" "20 5"; do set -- $K
    python bench.py --cpu-seconds 0 --verify 0 --isolated 0 --steps $1 --warmup $2 --waves $w --thresh $t 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], end='  ')"
  done; echo
done; done
