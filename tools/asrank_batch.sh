cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6t
B="--cpu-seconds 0 --moving 0 --default-abi 0 --long-steps 0 --isolated 0"
for n in 8 4; do for b in 4 8 16 32; do
  v=$(python bench.py --as-rank 0/$n --batch $b $B --steps 640 --warmup 64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['verified'])")
  echo "as-rank 0/$n batch $b: $v"; done; done > gpurun_out/r6t/asrank_batch.txt
cat gpurun_out/r6t/asrank_batch.txt
