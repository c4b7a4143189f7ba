#!/bin/bash
# kernel timeline of a short what-if run (rank 0's share at N = 8, --steps 20 --warmup 5): start / end of every persistent launch
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trace_short; rm -rf $O
for i in 1 2 3; do
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $O/$i -- python3 $GRAFT_REPO_ROOT/bench.py --as-rank 0/8 --steps 20 --warmup 5 --cpu-seconds 0 --verify 0 --isolated 0 2>/dev/null | tail -1 | cut -c1-120
python3 - <<PY
import csv,glob
f=glob.glob("$O/$i/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "persist_kernel" in r["Kernel_Name"]]
t0=int(rows[0]["Start_Timestamp"])
for r in rows: print("  start %8.3f ms  dur %7.3f ms  queue %s" % ((int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r.get("Queue_Id")))
PY
done
