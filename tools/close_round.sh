#!/bin/bash
# Closing run of a build:  bash tools/close_round.sh <tag, e.g. round4>   (on the GPU box, from the repository root)
#   whole GPU suite, PMC passes of the default bench command and of one frame at a time (-> profiles/pmc_per_launch.json, hash-
#   stamped), bench lines of every configuration / camera path / what-if rank / driver, the config table, rocprofv3 kernel
#   traces of the default, the driver's and the one-frame command.  Everything lands in gpurun_out/close_<tag>/; what is
#   committed is copied to profiles/<tag>_* by the caller.
cd $GRAFT_REPO_ROOT
TAG=${1:-round}
O=gpurun_out/close_$TAG; mkdir -p $O
line() { tail -1 | cut -c1-100000; }
python -m pytest tests -q -m gpu 2>&1 | tail -2 > $O/gpu_suite.txt; cat $O/gpu_suite.txt
python tools/pmc_pass.py --tag $TAG -- > $O/pmc_pass.log 2>&1; tail -1 $O/pmc_pass.log | cut -c1-200
python tools/pmc_pass.py --tag $TAG -- --inflight 1 --batch 1 > $O/pmc_pass1.log 2>&1; tail -1 $O/pmc_pass1.log | cut -c1-200
cp gpurun_out/pmc_per_launch.json profiles/pmc_per_launch.json
# lanes traversing per trip (an -DSVO_STAMPS=1 build of the same sources: make -C svo-raytracer_amd/csrc variant VARIANT=stamps EXTRA=-DSVO_STAMPS=1)
if [ -f svo-raytracer_amd/csrc/libsvohip_stamps.so ]; then
  STAMPS_JSON=1 STAMPS_WAVES=10 STAMPS_BATCH=4 SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_stamps.so python tools/stamps.py > $O/stamps.txt 2>&1
  STAMPS_WAVES=10,24 STAMPS_BATCH=1 SVO_HIP_LIB=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_stamps.so python tools/stamps.py >> $O/stamps.txt 2>&1
  cp gpurun_out/stamps_per_launch.json profiles/stamps_per_launch.json; tail -3 $O/stamps.txt | cut -c1-250
fi
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | line > $O/bench_driver_k20.json
B="--cpu-seconds 0 --moving 0 --default-abi 0 --long-steps 0"
V=$GRAFT_REPO_ROOT/svo-raytracer_amd/csrc/libsvohip_variants.so   # the comparators and the SVO_* switches live there
python bench.py --inflight 1 --batch 1 $B 2>/dev/null | line > $O/bench_inflight1.json
python bench.py --camera-path orbit $B 2>/dev/null | line > $O/bench_orbit.json
python bench.py --camera-path orbit --inflight 1 --batch 1 $B 2>/dev/null | line > $O/bench_orbit_inflight1.json
SVO_HIP_LIB=$V SVO_DERIVED=0 python bench.py $B 2>/dev/null | line > $O/bench_recordwalk.json
SVO_HIP_LIB=$V SVO_SPARE=1 python bench.py $B 2>/dev/null | line > $O/bench_spare_kernel.json
SVO_HIP_LIB=$V SVO_SPARE=1 python bench.py --inflight 1 --batch 1 $B 2>/dev/null | line > $O/bench_spare_kernel_inflight1.json
python bench.py --beam 1 $B 2>/dev/null | line > $O/bench_beam1.json
python bench.py --mode 2 $B 2>/dev/null | line > $O/bench_mode2.json
python bench.py --config C2 $B 2>/dev/null | line > $O/bench_C2.json
python bench.py --config C4 $B --steps 60 2>/dev/null | line > $O/bench_C4.json
python bench.py --config C5 $B --steps 12 --warmup 2 2>/dev/null | line > $O/bench_C5.json
python bench.py --config C5spp $B --steps 12 --warmup 2 2>/dev/null | line > $O/bench_C5spp.json
for n in 2 4 8; do python bench.py --as-rank 0/$n $B --steps 400 --warmup 40 2>/dev/null | line > $O/bench_asrank0of$n.json; done
python bench.py --config C4 --as-rank 0/8 $B --steps 120 2>/dev/null | line > $O/bench_C4_asrank0of8.json
python bench.py --config C5 --as-rank 0/8 $B --steps 12 --warmup 3 2>/dev/null | line > $O/bench_C5_asrank0of8.json
SVO_BENCH_FORCE_COMM=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 $B 2>/dev/null | line > $O/bench_forcecomm_world1.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | line > $O/bench_torchrun_world1_k20.json
SVO_BENCH_BACKEND=gloo SVO_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --exchange copy --waves 5 $B --isolated 0 2>/dev/null | line > $O/bench_two_ranks_one_gpu_copy.json
for n in 2 8; do SVO_BENCH_ONE_GPU=1 python bench.py --gpus $n --driver group --exchange copy $B --steps 100 --warmup 12 2>/dev/null | line > $O/bench_group${n}_one_gpu.json; done
python tests/config_table.py > $O/config_table.md 2>&1; cat $O/config_table.md
# round 6: the measurement matrix (cameras K0 / K1 / K2 x three scenes) and the reference's loop by launch shape
python tools/matrix.py --out $O/matrix > $O/matrix.log 2>&1; tail -12 $O/matrix.log | cut -c1-200
python tools/loop_shape.py 0 24 16 12 10 > $O/loop_shape.log 2>&1; cp gpurun_out/loop_shape.txt $O/loop_shape.txt
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 12 --cpu-seconds 0 --verify 0 --isolated 0 --moving 0 --default-abi 0 --long-steps 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_k20 -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --verify 0 --moving 0 --default-abi 0 --long-steps 0 > $GRAFT_REPO_ROOT/$O/trace_k20.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 --verify 0 --inflight 1 --batch 1 --moving 0 --default-abi 0 --long-steps 0 > $GRAFT_REPO_ROOT/$O/trace1.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_c5 -- python3 $GRAFT_REPO_ROOT/bench.py --config C5 --steps 8 --warmup 2 --cpu-seconds 0 --verify 0 --isolated 0 > $GRAFT_REPO_ROOT/$O/trace_c5.log 2>&1
cd $GRAFT_REPO_ROOT
for t in trace trace_k20 trace1 trace_c5; do python tools/pmc_summary.py $O/$t > $O/${t}_summary.txt 2>&1; head -5 $O/${t}_summary.txt; tail -1 $O/$t.log | cut -c1-300 >> $O/${t}_summary.txt; rm -rf $O/$t; done
rm -rf gpurun_out/pmc_${TAG}_*
python - <<PY
import json,glob
for n in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.loads(open(n).read().strip().splitlines()[-1])
        print(n.split("/")[-1], j["value"], j["ms_per_step"], j["verified"], j["roofline"]["frac"], j["roofline"]["kernel_ms"], j["roofline"].get("traffic"), j.get("value_one_frame_at_a_time"), j.get("value_moving_camera"), j.get("value_long_run"))
    except Exception as e:
        print(n, "unreadable", e)
PY
