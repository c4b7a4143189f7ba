#!/bin/bash
run() { python bench.py --steps 100 --warmup 6 --cpu-seconds 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
echo -n "base thresh=5: "; SVO_HIP_LIB=$PWD/svo-raytracer_amd/csrc/libsvohip_base.so SVO_PERSIST_THRESH=5 run
for t in 3 4 5 6 7 8 10; do echo -n "polB free>=$((t*4)): "; SVO_HIP_LIB=$PWD/svo-raytracer_amd/csrc/libsvohip_polB.so SVO_PERSIST_THRESH=$t run; done
