# shader / memory clocks and socket power sampled with rocm-smi while a bounded bench run keeps the GPU busy
cd $GRAFT_REPO_ROOT
timeout 120 python bench.py --steps 40000 --warmup 24 --cpu-seconds 0 --moving 0 --default-abi 0 --long-steps 0 --verify 0 --isolated 0 > /tmp/b.log 2>&1 &
pid=$!
sleep 8
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power|mclk|fclk" | head -6; echo --; sleep 0.5; done
wait $pid; tail -1 /tmp/b.log | cut -c1-200
rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -2
