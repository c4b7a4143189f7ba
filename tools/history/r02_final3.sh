#!/bin/bash
# closing numbers on the final sources: bench lines, PMC, traces
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_final3; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --inflight 1 --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_inflight1.json
python bench.py --mode 2 --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_mode2.json
python bench.py --config C2 --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_C2.json
python bench.py --config C4 --cpu-seconds 0 --steps 60 2>/dev/null | tail -1 > $O/bench_C4.json
python bench.py --config C5 --cpu-seconds 0 --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/bench_C5.json
python tools/pmc_pass.py --tag r02e -- > $O/pmc_pass.log 2>&1; tail -2 $O/pmc_pass.log
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 --verify 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 --verify 0 --inflight 1 > $GRAFT_REPO_ROOT/$O/trace1.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/trace > $O/trace_summary.txt 2>&1; head -4 $O/trace_summary.txt
python tools/pmc_summary.py $O/trace1 > $O/trace1_summary.txt 2>&1; head -4 $O/trace1_summary.txt
python - <<PY
import json
for n in ("bench_default","bench_inflight1","bench_mode2","bench_C2","bench_C4","bench_C5"):
    j=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1])
    print(n, j["value"], j["ms_per_step"], j["verified"], j["roofline"]["frac"], j["config"]["rays_per_frame"], j["config"]["iterations_per_ray"])
PY
