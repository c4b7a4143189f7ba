cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
python -m pytest tests/test_gpu_pick.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r6b/pick_tests.txt
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_pick.py 2>&1 | tail -15 > gpurun_out/r6b/gpu_suite.txt
python bench.py > gpurun_out/r6b/bench_default.json 2> gpurun_out/r6b/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6b/bench_k20.json 2> gpurun_out/r6b/bench_k20.err
python tools/matrix.py --out gpurun_out/r6b/matrix > gpurun_out/r6b/matrix.log 2>&1
tail -3 gpurun_out/r6b/pick_tests.txt; tail -3 gpurun_out/r6b/gpu_suite.txt
