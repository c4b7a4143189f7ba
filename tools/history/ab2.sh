#!/bin/bash
# A/B of library variants for pipelines 1 and 2; usage: tools/ab2.sh name1 name2 ...
run() { python bench.py --steps 100 --warmup 6 --cpu-seconds 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for v in "$@"; do
  export SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_$v.so
  for w in 10 12; do echo -n "$v p1 waves/cu=$w: "; SVO_PERSIST_WAVES_PER_CU=$w run --pipeline 1; done
  for w in 0 12; do echo -n "$v p2 waves/cu=$w thresh5: "; SVO_WF_WAVES_PER_CU=$w SVO_WF_THRESH=5 run --pipeline 2; done
done
