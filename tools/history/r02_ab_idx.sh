#!/bin/bash
# A/B: child index derived from the position bits each trip (idx1, default) vs carried and rebuilt (idx0)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q -m gpu -k "asm_loop or golden or iteration_cap or config3" 2>&1 | tail -3
bash tools/ab.sh "--steps 400 --verify 0" idx0 idx1 idx0 idx1
bash tools/ab.sh "--steps 400 --verify 0 --inflight 1" idx0 idx1
