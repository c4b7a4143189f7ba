#!/bin/bash
cd $GRAFT_REPO_ROOT
SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_stamps.so python tools/r03_timeline.py 2>&1 | grep -v amdgpu.ids
echo "--- upper bound of any output staging: the kernel without its colour / depth stores (SVO_NO_STORES build, no verification possible)"
for r in 1 2; do for v in "" _nostores; do
  echo -n "libsvohip$v.so r$r: "
  SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip$v.so python bench.py --steps 400 --verify 0 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
