#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_comm.txt; : > $O
run() { echo -n "$1 | $2: " >> $O; env $1 SVO_BENCH_FORCE_COMM=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --cpu-seconds 0 $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" >> $O 2>&1; }
run "SVO_RING_HOST_WAIT=1" ""
run "SVO_RING_HOST_WAIT=0" ""
run "SVO_RING_HOST_WAIT=1" "--waves 6"
run "SVO_RING_HOST_WAIT=0" "--waves 6"
run "SVO_RING_HOST_WAIT=1" "--inflight 6"
run "SVO_RING_HOST_WAIT=0" "--inflight 6"
run "SVO_RING_HOST_WAIT=1" "--comm-cus 0"
run "SVO_RING_HOST_WAIT=1" "--comm-cus 1"
run "SVO_RING_HOST_WAIT=1" "--comm-cus 2"
run "SVO_RING_HOST_WAIT=1" "--comm-cus 2 --inflight 6"
run "SVO_RING_HOST_WAIT=1" "--comm-cus 1 --inflight 6"
echo -n "no comm: " >> $O; timeout 600 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" >> $O 2>&1
cat $O
