#!/bin/bash
# kernel trace of the default bench command without the single-frame launches of kernel_ms_isolated: every launch of the
# dominant kernel is a timed-region (or warm-up) launch of 4 frames, so the trace's average is comparable with kernel_ms
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_trace; mkdir -p $O
python -m pytest tests/test_gpu_inflight.py -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 12 --cpu-seconds 0 --verify 0 --isolated 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $O/trace.log > $O/bench_line.json
python tools/pmc_summary.py $O/trace > $O/trace_summary.txt 2>&1; head -5 $O/trace_summary.txt
python -c "
import json; j=json.loads(open('$O/bench_line.json').read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'])"
