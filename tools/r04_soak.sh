#!/bin/bash
# flake hunt on the final build: the GPU suite three times, then verified bench lines with the step / warm-up counts a
# driver might pass
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_soak.txt; : > $O
for i in 1 2 3; do python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -1 >> $O; done
for sw in "20 5" "10 2" "50 0" "7 3" "100 10" "1 0"; do set -- $sw
  echo -n "steps $1 warmup $2: " >> $O
  python bench.py --gpus 1 --steps $1 --warmup $2 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'], d['value_one_frame_at_a_time'])" >> $O 2>&1
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('under torch.distributed.run, N=1:', d['value'], d['ms_per_step'], d['verified'], d['ranks_seen'])" >> $O 2>&1
cat $O
