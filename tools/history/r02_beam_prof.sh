#!/bin/bash
# beam pre-pass: parity tests, A/B bench lines, kernel trace of a --beam 1 run
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_beam; mkdir -p $O
python -m pytest tests/test_beam.py -x -q -m gpu 2>&1 | tail -3
for b in 0 1; do python bench.py --beam $b --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_beam$b.json; python -c "
import json; j=json.load(open('$O/bench_beam$b.json')); print('beam $b:', j['value'], j['ms_per_step'], j['verified'], j['config']['iterations_per_ray'], j['roofline']['kernel_ms_isolated'])"; done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --beam 1 --steps 100 --warmup 5 --cpu-seconds 0 --verify 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/trace 2>&1 | head -8
