#!/bin/bash
# repeat the GPU suite and the verified bench line: races in the frames-in-flight logic would show as flakes
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -1; done
for i in 1 2 3 4 5 6; do python bench.py --cpu-seconds 0 --steps 203 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['verified'], d['config']['verification'][:60])"; done
for i in 1 2 3; do python bench.py --cpu-seconds 0 --beam 1 --steps 101 --inflight 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam', d['value'], d['verified'])"; done
python bench.py --cpu-seconds 0 --config C5 --steps 5 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5', d['value'], d['verified'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
