#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_inflight.py tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_accum.py -x -q -m gpu 2>&1 | tail -3
for fb in 0 4294967296; do for r in 1 2; do
  echo -n "C5 SVO_FOLD_BYTES=$fb r$r: "
  SVO_FOLD_BYTES=$fb python bench.py --config C5 --cpu-seconds 0 --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['config']['frames_per_launch'])"
done; done
for fb in 0 4294967296; do
  echo -n "C3 --spp 4 SVO_FOLD_BYTES=$fb: "
  SVO_FOLD_BYTES=$fb python bench.py --spp 4 --cpu-seconds 0 --steps 60 --warmup 6 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])"
  echo -n "C5 --as-rank 0/8 SVO_FOLD_BYTES=$fb: "
  SVO_FOLD_BYTES=$fb python bench.py --config C5 --as-rank 0/8 --cpu-seconds 0 --steps 12 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])"
done
bash tools/history/r03_fold_trace.sh
