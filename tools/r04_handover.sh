#!/bin/bash
# hand-over of a draining launch's last paths: parity with it on, then single-frame and throughput A/B by levels x threshold
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_handover.txt; : > $O
for cfg in "2 20" "3 32" "1 8"; do set -- $cfg
  echo "== parity, SVO_HANDOVER=$1 SVO_HANDOVER_BELOW=$2" >> $O
  SVO_HANDOVER=$1 SVO_HANDOVER_BELOW=$2 timeout 900 python -m pytest tests/test_gpu_derived.py tests/test_gpu_parity.py tests/test_gpu_inflight.py tests/test_config3.py tests/test_accum.py -x -q -m gpu 2>&1 | tail -3 >> $O
done
for args in "--steps 100 --warmup 10 --inflight 1 --batch 1" "--steps 200 --warmup 12"; do
  for cfg in "0 20" "1 16" "1 24" "2 16" "2 24" "2 32" "3 24" "3 32" "0 20"; do set -- $cfg
    echo -n "levels $1 below $2 | $args: " >> $O
    SVO_HANDOVER=$1 SVO_HANDOVER_BELOW=$2 timeout 600 python bench.py $args --cpu-seconds 0 --isolated 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('verified'))" >> $O 2>&1
  done
done
cat $O
