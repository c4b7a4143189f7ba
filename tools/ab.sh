#!/bin/bash
# A/B several builds of libsvohip.so (made beforehand with `make VARIANT=name EXTRA="-D..."`) in one GPU call,
# interleaved rounds.  usage: tools/ab.sh "<bench args>" name1 name2 ...
ARGS=$1; shift
for round in 1 2; do
  for v in "$@"; do
    echo -n "$v r$round: "
    SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_$v.so python bench.py $ARGS --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
