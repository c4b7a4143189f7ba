#!/bin/bash
# A/B: child-block prefetch in the descend section (pf1, default) vs none (pf0)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_inflight.py -x -q -m gpu -k "asm_loop or golden or iteration_cap or config3 or config4 or bench_configuration" 2>&1 | tail -3
bash tools/ab.sh "--steps 400 --verify 0" pf0 pf1 pf0 pf1
bash tools/ab.sh "--steps 400 --verify 0 --inflight 1" pf0 pf1
