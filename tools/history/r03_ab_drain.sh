#!/bin/bash
# A/B: refill threshold once every band is used up (d<N> = N/16 of the active lanes may still traverse when a round starts;
# d0 = wait for all) x next band in the same round (s1) or one band per round (s0)
cd $GRAFT_REPO_ROOT
V="d0s0 d12s0 d16s0 d9s1 d12s1 d16s1"
bash tools/ab.sh "--steps 400 --verify 0 --inflight 1 --batch 1" $V
bash tools/ab.sh "--steps 400 --verify 0" $V
bash tools/ab.sh "--steps 400 --verify 0 --as-rank 0/8" $V
