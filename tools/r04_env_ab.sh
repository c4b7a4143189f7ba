#!/bin/bash
# A/B of environment switches on the current build: tools/r04_env_ab.sh "VAR=a" "VAR=b" ...   (interleaved, 3 rounds)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_env_ab.txt; : > $O
for args in "--steps 100 --warmup 10" "--steps 100 --warmup 10 --inflight 1 --batch 1"; do
  for r in 1 2 3; do
    for v in "$@"; do
      echo -n "$v | $args: " >> $O
      env $v timeout 600 python bench.py $args --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('verified'))" >> $O 2>&1
    done
  done
done
cat $O
