#!/bin/bash
# kernel trace of the driver's own bench invocation
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03_driver_trace; rm -rf $O; mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2>&1
cd $GRAFT_REPO_ROOT
grep '^{"metric"' $O/bench.log | tail -1 > $O/bench_line.json
python tools/pmc_summary.py $O/t > $O/summary.txt 2>&1; head -14 $O/summary.txt; rm -rf $O/t
python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['value'], d['ms_per_step'], d['verified'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_isolated'])"
