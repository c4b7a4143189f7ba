cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d; mkdir -p $O
python -m pytest tests/test_gpu_pick.py -m gpu -q -x 2>&1 | tail -15 > $O/pick_tests.txt
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_pick.py 2>&1 | tail -15 > $O/gpu_suite.txt
python tools/loop_shape.py > $O/loop_shape.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
tail -3 $O/pick_tests.txt; tail -3 $O/gpu_suite.txt; tail -9 $O/loop_shape.log
