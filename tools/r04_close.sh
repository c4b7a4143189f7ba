#!/bin/bash
# closing run of round 3's build: whole GPU suite, PMC passes (default + one frame at a time), bench lines of every config
# and what-if rank, config table, kernel traces -> gpurun_out/r04_close/ (copied to profiles/round3_* afterwards)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_close; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -2 > $O/gpu_suite.txt; cat $O/gpu_suite.txt
python tools/pmc_pass.py --tag r04 -- > $O/pmc_pass.log 2>&1; tail -1 $O/pmc_pass.log | cut -c1-200
python tools/pmc_pass.py --tag r04 -- --inflight 1 --batch 1 > $O/pmc_pass1.log 2>&1; tail -1 $O/pmc_pass1.log | cut -c1-200
cp gpurun_out/pmc_per_launch.json profiles/pmc_per_launch.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_k20.json
python bench.py --inflight 1 --batch 1 --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_inflight1.json
SVO_DERIVED=0 python bench.py --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_recordwalk.json
python bench.py --beam 1 --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_beam1.json
python bench.py --mode 2 --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_mode2.json
python bench.py --config C2 --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_C2.json
python bench.py --config C4 --cpu-seconds 0 --steps 60 2>/dev/null | tail -1 > $O/bench_C4.json
python bench.py --config C5 --cpu-seconds 0 --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/bench_C5.json
for n in 2 4 8; do python bench.py --as-rank 0/$n --cpu-seconds 0 --steps 400 --warmup 40 2>/dev/null | tail -1 > $O/bench_asrank0of$n.json; done
python bench.py --config C4 --as-rank 0/8 --cpu-seconds 0 --steps 120 2>/dev/null | tail -1 > $O/bench_C4_asrank0of8.json
python bench.py --config C5 --as-rank 0/8 --cpu-seconds 0 --steps 12 --warmup 3 2>/dev/null | tail -1 > $O/bench_C5_asrank0of8.json
SVO_BENCH_FORCE_COMM=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --cpu-seconds 0 2>/dev/null | tail -1 > $O/bench_forcecomm_world1.json
SVO_BENCH_BACKEND=gloo SVO_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --exchange copy --waves 5 --cpu-seconds 0 --isolated 0 2>/dev/null | tail -1 > $O/bench_two_ranks_one_gpu_copy.json
python tests/config_table.py > $O/config_table.md 2>&1; cat $O/config_table.md
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 12 --cpu-seconds 0 --verify 0 --isolated 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_k20 -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --verify 0 > $GRAFT_REPO_ROOT/$O/trace_k20.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 --verify 0 --inflight 1 --batch 1 > $GRAFT_REPO_ROOT/$O/trace1.log 2>&1
cd $GRAFT_REPO_ROOT
for t in trace trace_k20 trace1; do python tools/pmc_summary.py $O/$t > $O/${t}_summary.txt 2>&1; head -5 $O/${t}_summary.txt; tail -1 $O/$t.log | cut -c1-300 >> $O/${t}_summary.txt; rm -rf $O/$t; done
rm -rf gpurun_out/pmc_r04_*
python - <<PY
import json,glob
for n in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.loads(open(n).read().strip().splitlines()[-1])
        print(n.split("/")[-1], j["value"], j["ms_per_step"], j["verified"], j["roofline"]["frac"], j["roofline"]["kernel_ms"], j["roofline"].get("traffic"), j.get("value_one_frame_at_a_time"))
    except Exception as e:
        print(n, "unreadable", e)
PY
