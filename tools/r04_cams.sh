cd $GRAFT_REPO_ROOT
for cam in K0 K2; do for mode in 0 2; do python bench.py --camera $cam --mode $mode --steps 60 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cam', $mode, d['value'], d['verified'], d.get('verify_info'))"; done; done
python bench.py --config C4 --steps 12 --warmup 2 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4', d['value'], d['verified'])"
python bench.py --hits 1 --beam 1 --steps 40 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hits+beam', d['value'], d['verified'])"
