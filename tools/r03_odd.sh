#!/bin/bash
# bench.py with the step / warm-up counts a driver might pass
cd $GRAFT_REPO_ROOT
for a in "--steps 1 --warmup 0" "--steps 2 --warmup 1" "--steps 3 --warmup 0" "--steps 5 --warmup 5" "--steps 7 --warmup 2" "--steps 20 --warmup 3" "--steps 50 --warmup 10 --gpus 1" "--steps 9 --warmup 1 --inflight 1" "--steps 10 --warmup 2 --batch 8" "--steps 4 --warmup 0 --config C4" "--steps 2 --warmup 0 --config C5" "--steps 11 --warmup 3 --mode 2" "--steps 6 --warmup 1 --config C2" "--steps 13 --warmup 2 --pipeline 2" "--steps 13 --warmup 2 --pipeline 0"; do
  echo -n "$a -> "
  python bench.py --cpu-seconds 0 $a 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['verified'], d['n_gpus'])
except Exception as e: print('FAILED', e)"
done
