#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_inflight.py tests/test_gpu_parity.py tests/test_accum.py -x -q -m gpu 2>&1 | tail -2
for cfg in "--config C5 --steps 6 --warmup 2" "--config C5 --steps 6 --warmup 2 --inflight 1" "--spp 4 --steps 40" "--steps 200"; do
python bench.py $cfg --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$cfg', j['value'], j['ms_per_step'], j['verified'])"; done
