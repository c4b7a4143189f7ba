#!/bin/bash
cd $GRAFT_REPO_ROOT
echo -n "C5 --batch 1 one launch per sample: "
SVO_FOLD_BYTES=0 python bench.py --config C5 --batch 1 --cpu-seconds 0 --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])"
for g in 1 8 64 512 100000; do
  echo -n "C5 --batch 1 SVO_FOLD_GROUP=$g: "
  SVO_FOLD_GROUP=$g python bench.py --config C5 --batch 1 --cpu-seconds 0 --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])"
done
