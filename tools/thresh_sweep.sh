cd $GRAFT_REPO_ROOT; export SVO_SCENE_CACHE=/tmp/svo_scene_cache; mkdir -p $SVO_SCENE_CACHE gpurun_out/r6p
for cam in K0 K1; do for t in 5 6 7 8 9 10 11; do
  v=$(python bench.py --scene dust --camera $cam --thresh $t --steps 400 --warmup 24 --long-steps 0 --moving 0 --default-abi 0 --by-camera 0 --ref-loop 0 --cpu-seconds 0 --isolated 0 --verify 0 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  echo "dust $cam thresh $t $v"; done; done > gpurun_out/r6p/thresh.txt
for t in 7 8 9 10; do
  v=$(python bench.py --thresh $t --steps 400 --warmup 24 --long-steps 0 --moving 0 --default-abi 0 --by-camera 0 --ref-loop 0 --cpu-seconds 0 --isolated 0 --verify 0 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  echo "terrain K1 thresh $t $v"; done >> gpurun_out/r6p/thresh.txt
cat gpurun_out/r6p/thresh.txt
