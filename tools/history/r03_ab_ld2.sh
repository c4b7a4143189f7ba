#!/bin/bash
# A/B: second dword of the child record for every active lane (ld2all) vs only for lanes that can descend (ld2narrow)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_gpu_edge.py -x -q -m gpu 2>&1 | tail -3
bash tools/ab.sh "--steps 400 --verify 0" ld2all ld2narrow
bash tools/ab.sh "--steps 400 --verify 0 --inflight 1 --batch 1" ld2all ld2narrow
