#!/bin/bash
# A/B: the child record as two dword loads of any alignment (two) vs one aligned dwordx3 + 2 v_alignbyte_b32 (x3)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_gpu_edge.py -x -q -m gpu 2>&1 | tail -3
bash tools/ab.sh "--steps 400 --verify 0" two x3
bash tools/ab.sh "--steps 400 --verify 0 --inflight 1 --batch 1" two x3
