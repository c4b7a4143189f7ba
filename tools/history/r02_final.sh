#!/bin/bash
# round-2 final measurement session: full GPU suite, default + inflight-1 bench lines, PMC passes, kernel trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_final; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err
python bench.py --inflight 1 --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_inflight1.json
python tools/pmc_pass.py --tag r02c -- > $O/pmc_pass.log 2>&1; tail -3 $O/pmc_pass.log
python bench.py --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_with_pmc.json
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 --verify 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 --verify 0 --inflight 1 > $GRAFT_REPO_ROOT/$O/trace1.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/trace > $O/trace_summary.txt 2>&1; head -6 $O/trace_summary.txt
python tools/pmc_summary.py $O/trace1 > $O/trace1_summary.txt 2>&1; head -4 $O/trace1_summary.txt
python - <<PY
import json
for n in ("bench_default","bench_inflight1","bench_with_pmc"):
    j=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1])
    print(n, j["value"], j["ms_per_step"], j["verified"], j["roofline"]["frac"], j["roofline"]["traffic"], j["roofline"].get("binding_roof"), j["config"]["scene_build_s"])
PY
