#!/bin/bash
# instruction-mix counters of the persistent kernel for two library variants (one frame at a time)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  OUT=gpurun_out/pmc_ab_$v; rm -rf $OUT
  SVO_HIP_LIB=$PWD/svo-raytracer_amd/csrc/libsvohip_$v.so timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 bench.py --steps 30 --warmup 3 --cpu-seconds 0 --inflight 1 > $OUT.log 2>&1
  echo "== $v"; python3 tools/pmc_summary.py $OUT 2>&1 | grep -E "persist_kernel" | awk '{print $3, $5}' | tr '\n' ' '; echo
done
