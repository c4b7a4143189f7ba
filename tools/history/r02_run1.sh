#!/bin/bash
# round-2 GPU session 1: RCCL path at world 1, launcher rejection, strong-scaling what-ifs, PMC + kernel-trace passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_run1; mkdir -p $O
SVO_BENCH_FORCE_COMM=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 5 --cpu-seconds 0 > $O/forcecomm.json 2> $O/forcecomm.err; echo "forcecomm rc $?"; tail -c 600 $O/forcecomm.json
python bench.py --gpus 2 --steps 5 > $O/gpus2.out 2>&1; echo "gpus2 rc $?"; tail -2 $O/gpus2.out
for spec in "0/2 3" "0/4 3" "0/8 3" "0/8 8" "3/8 8" "0/4 6"; do
  set -- $spec
  python bench.py --as-rank $1 --inflight $2 --cpu-seconds 0 --steps 300 > $O/asrank_${1//\//of}_if$2.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("$O/asrank_${1//\//of}_if$2.json").read().strip().splitlines()[-1])
print("as-rank $1 inflight $2: ms/step %.4f rays/frame %d value %.1f verified %s" % (j["ms_per_step"], j["config"]["rays_per_frame"], j["value"], j["verified"]))
PY
done
python tools/pmc_pass.py --tag r02a -- > $O/pmc_pass.log 2>&1; tail -25 $O/pmc_pass.log
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/trace > $O/trace_summary.txt 2>&1; head -20 $O/trace_summary.txt
