#!/bin/bash
# interleaved A/B of environment switches on the current build: ROUNDS (default 6) passes over the settings, one bench run each
# (BENCH_ARGS, default 400 steps of the default configuration; --moving 0), then mean / min / max per setting.
#   usage: tools/env_ab.sh "VAR=a" "VAR=b" ...        ("-" = no variable set)
cd $GRAFT_REPO_ROOT
O=gpurun_out/env_ab.txt; : > $O
for r in $(seq 1 ${ROUNDS:-6}); do
  for v in "$@"; do
    echo -n "$v " >> $O
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    env $e timeout 600 python bench.py ${BENCH_ARGS:---steps 400 --warmup 24} --cpu-seconds 0 --moving 0 --default-abi 0 --long-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['verified'])" >> $O 2>&1
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
        s="%-24s n=%d mean %.1f min %.1f max %.1f" % (k,len(v),sum(v)/len(v),min(v),max(v)); print(s); f.write(s+"\n")
PY
