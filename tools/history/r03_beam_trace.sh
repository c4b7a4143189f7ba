#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03_beam_trace; rm -rf $O; mkdir -p $O
for cam in K1 K0; do
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$cam -- python3 $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 12 --cpu-seconds 0 --verify 0 --isolated 0 --beam 1 --camera $cam > $O/$cam.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${cam}_1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 6 --cpu-seconds 0 --verify 0 --isolated 0 --beam 1 --camera $cam --inflight 1 --batch 1 > $O/${cam}_1.log 2>&1
done
cd $GRAFT_REPO_ROOT
for d in K1 K1_1 K0 K0_1; do echo "== $d"; python tools/pmc_summary.py $O/$d | grep -E "persist_kernel|beam_kernel" | cut -c1-160; rm -rf $O/$d; done
