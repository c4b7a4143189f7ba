#!/bin/bash
# interleaved A/B of library variants in the throughput configuration: ROUNDS (default 8) passes over "base v1 v2 ...",
# one bench run each (400 steps), then mean / min / max per variant -- for effects of a per cent, which the three-in-a-row
# runs of r04_ab.sh cannot separate from the drift of a box
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_interleaved.txt; : > $O
for r in $(seq 1 ${ROUNDS:-8}); do
  for v in "" "$@"; do
    lib=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip${v:+_$v}.so
    echo -n "${v:-base} " >> $O
    SVO_HIP_LIB=$lib timeout 600 python bench.py ${BENCH_ARGS:---steps 400 --warmup 24} --cpu-seconds 0 --moving 0 --default-abi 0 --long-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['verified'])" >> $O 2>&1
  done
done
python - <<PY
import collections
d=collections.defaultdict(list)
for l in open("$O"):
    p=l.split()
    try: d[p[0]].append(float(p[1]))
    except Exception: pass
with open("$O","a") as f:
    for k,v in d.items():
        s="%-10s n=%d mean %.1f min %.1f max %.1f" % (k,len(v),sum(v)/len(v),min(v),max(v)); print(s); f.write(s+"\n")
PY
