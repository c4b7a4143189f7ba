#!/bin/bash
# interleaved A/B of bench.py argument sets (one quoted string each), ROUNDS passes, 400 steps: args_ab.sh "--thresh 9" "--thresh 8" ...
cd $GRAFT_REPO_ROOT
O=gpurun_out/args_ab.txt; : > $O
for r in $(seq 1 ${ROUNDS:-6}); do
  for a in "$@"; do
    echo -n "[$a] " >> $O
    timeout 600 python bench.py --steps 400 --warmup 24 --cpu-seconds 0 --isolated 0 --moving 0 --default-abi 0 --long-steps 0 $a 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['verified'])" >> $O 2>&1
  done
done
python - <<PY
import collections,re
d=collections.defaultdict(list)
for l in open("$O"):
    m=re.match(r"\[(.*)\] ([0-9.]+) ", l)
    if m: d[m.group(1)].append(float(m.group(2)))
with open("$O","a") as f:
    for k,v in d.items():
        s="%-28s n=%d mean %.1f min %.1f max %.1f" % (k,len(v),sum(v)/len(v),min(v),max(v)); print(s); f.write(s+"\n")
PY
