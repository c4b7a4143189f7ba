#!/bin/bash
# quick parity subset + the two headline bench lines of the current build (and of variants named on the command line)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab.txt; : > $O
if [ -z "$SKIPTESTS" ]; then timeout 1200 python -m pytest tests/test_gpu_derived.py tests/test_gpu_parity.py tests/test_gpu_inflight.py tests/test_accum.py tests/test_beam.py tests/test_gpu_edge.py -x -q -m gpu 2>&1 | tail -4 >> $O; fi
for v in "" "$@"; do
  lib=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip${v:+_$v}.so
  for args in "--steps 200 --warmup 12" "--steps 100 --warmup 10 --inflight 1 --batch 1"; do
    for r in 1 2 3; do
      echo -n "${v:-base} $args: " >> $O
      SVO_HIP_LIB=$lib timeout 600 python bench.py $args --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('verified'))" >> $O 2>&1
    done
  done
done
cat $O
