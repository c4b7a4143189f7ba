#!/bin/bash
# after a change to the kernel sources: smoke, the PMC passes and the three headline lines again (hash-stamped)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_close; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python tools/pmc_pass.py --tag r04 -- > $O/pmc_pass.log 2>&1; tail -1 $O/pmc_pass.log | cut -c1-200
python tools/pmc_pass.py --tag r04 -- --inflight 1 --batch 1 > $O/pmc_pass1.log 2>&1; tail -1 $O/pmc_pass1.log | cut -c1-200
cp gpurun_out/pmc_per_launch.json profiles/pmc_per_launch.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_k20.json
python bench.py --inflight 1 --batch 1 --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_inflight1.json
rm -rf gpurun_out/pmc_r04_*
python - <<PY
import json,glob
for n in ("bench_default","bench_driver_k20","bench_inflight1"):
    j=json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
    print(n, j["value"], j["ms_per_step"], j["verified"], j["roofline"]["frac"], j["roofline"].get("traffic"), j.get("value_one_frame_at_a_time"), j["config"].get("descriptor_table"))
PY
