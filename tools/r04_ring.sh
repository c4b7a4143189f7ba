#!/bin/bash
# round-3 session 3: the C-ABI frame ring -- tests, then bench.py driving it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ring.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_config3.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -15 >> $O
for args in "--steps 100 --warmup 10" "--steps 20 --warmup 5" "--steps 100 --warmup 10 --inflight 1 --batch 1" "--steps 60 --warmup 10 --as-rank 0/8"; do
  echo "== bench $args" >> $O
  timeout 600 python bench.py $args --cpu-seconds 0 2>&1 | tail -3 | cut -c1-1500 >> $O
done
echo "== forced comm at world size 1" >> $O
SVO_BENCH_FORCE_COMM=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --cpu-seconds 0 2>&1 | tail -2 | cut -c1-600 >> $O
cat $O
