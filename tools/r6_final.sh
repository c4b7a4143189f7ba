# closing passes on the final sources: the GPU suite, the PMC and stamps passes bench.py's line is hash-gated on, the two bench
# lines the driver's run corresponds to, and the rocprofv3 kernel trace of the default command
cd $GRAFT_REPO_ROOT
O=gpurun_out/final6; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/gpu_suite.txt; tail -2 $O/gpu_suite.txt
python tools/pmc_pass.py --tag round6 -- > $O/pmc_pass.log 2>&1; tail -1 $O/pmc_pass.log | cut -c1-200
python tools/pmc_pass.py --tag round6 -- --inflight 1 --batch 1 > $O/pmc_pass1.log 2>&1; tail -1 $O/pmc_pass1.log | cut -c1-200
cp gpurun_out/pmc_per_launch.json profiles/pmc_per_launch.json
rm -f gpurun_out/stamps_per_launch.json
STAMPS_JSON=1 STAMPS_WAVES=10 STAMPS_BATCH=4 SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_stamps.so python tools/stamps.py > $O/stamps.txt 2>&1
STAMPS_WAVES=10,24 STAMPS_BATCH=1 SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_stamps.so python tools/stamps.py >> $O/stamps.txt 2>&1
cp gpurun_out/stamps_per_launch.json profiles/stamps_per_launch.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_k20.json 2> $O/bench_driver_k20.err
python bench.py --inflight 1 --batch 1 --cpu-seconds 0 --moving 0 --default-abi 0 --long-steps 0 2>/dev/null | tail -1 > $O/bench_inflight1.json
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 12 --cpu-seconds 0 --verify 0 --isolated 0 --moving 0 --default-abi 0 --long-steps 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_loop -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --verify 0 --isolated 0 --moving 0 --default-abi 0 --long-steps 0 --ref-loop 1 > $GRAFT_REPO_ROOT/$O/trace_loop.log 2>&1
cd $GRAFT_REPO_ROOT
for t in trace trace_loop; do python tools/pmc_summary.py $O/$t > $O/${t}_summary.txt 2>&1; head -6 $O/${t}_summary.txt | cut -c1-220; tail -1 $O/$t.log | cut -c1-300 >> $O/${t}_summary.txt; rm -rf $O/$t; done
rm -rf gpurun_out/pmc_round6_*
