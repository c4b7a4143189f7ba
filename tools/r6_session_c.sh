cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/gpu_suite.txt
ROUNDS=6 bash tools/ab_interleaved.sh nopick > $O/ab_nopick.txt 2>&1
python tools/loop_shape.py > $O/loop_shape.log 2>&1
STAMPS_JSON=1 STAMPS_WAVES=10 STAMPS_BATCH=4 SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_stamps.so python tools/stamps.py > $O/stamps_default.txt 2>&1
STAMPS_WAVES=24 STAMPS_BATCH=1 SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_stamps.so python tools/stamps.py > $O/stamps_one_frame.txt 2>&1
python tools/matrix.py --out $O/matrix --sweep t2a18_K1 > $O/sweep.log 2>&1
tail -3 $O/gpu_suite.txt; tail -3 $O/ab_nopick.txt; tail -8 $O/loop_shape.log
