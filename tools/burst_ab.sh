cd $GRAFT_REPO_ROOT
O=gpurun_out/burst_ab.txt; : > $O
for r in 1 2 3 4 5 6 7 8; do
  for a in "" "--waves 11" "--waves 12" "--waves 9"; do
    echo -n "[$a] " >> $O
    timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --isolated 0 --moving 0 --default-abi 0 --long-steps 400 $a 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['value_long_run'], d['verified'])" >> $O 2>&1
  done
done
python - <<PY
import collections,re
d=collections.defaultdict(list); l2=collections.defaultdict(list)
for l in open("$O"):
    m=re.match(r"\[(.*)\] ([0-9.]+) ([0-9.]+) ", l)
    if m: d[m.group(1)].append(float(m.group(2))); l2[m.group(1)].append(float(m.group(3)))
for k,v in d.items():
    print("%-12s n=%d burst mean %.1f min %.1f max %.1f | long mean %.1f" % (k or "default",len(v),sum(v)/len(v),min(v),max(v),sum(l2[k])/len(l2[k])))
PY
