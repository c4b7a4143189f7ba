#!/bin/bash
# round-3 session 1: the interior-descriptor table -- parity first, then A/B against the record walk
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r04_first.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_derived.py -x -q -m gpu 2>&1 | tail -15 >> $O
echo "== cxxloop lib, derived tests" >> $O
SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_cxxloop.so timeout 900 python -m pytest tests/test_gpu_derived.py -x -q -m gpu 2>&1 | tail -8 >> $O
for args in "--steps 100 --warmup 10" "--steps 100 --warmup 10 --inflight 1 --batch 1"; do
  for d in 0 1; do
    echo -n "SVO_DERIVED=$d $args: " >> $O
    SVO_DERIVED=$d timeout 600 python bench.py $args --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('verified'))" >> $O 2>&1
  done
  for v in early w5; do
    echo -n "variant $v $args: " >> $O
    SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_$v.so timeout 600 python bench.py $args --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('verified'))" >> $O 2>&1
  done
done
cat $O
