#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03_fold_trace; rm -rf $O; mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $GRAFT_REPO_ROOT/bench.py --config C5 --steps 6 --warmup 2 --cpu-seconds 0 --verify 0 --isolated 0 > $O/t.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/t | grep -E "persist|resolve|Name" | cut -c1-200; rm -rf $O/t
