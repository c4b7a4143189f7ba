cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr && timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/bench.py ${TRACE_ARGS:---gpus 1 --steps 20 --warmup 5} --cpu-seconds 0 --verify 0 --moving 0 --isolated 0 > /tmp/tr.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/tr/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:60],r.get('Stream_Id') or r.get('Queue_Id')) for r in rows if 'persist_kernel' in r['Kernel_Name'] or 'fillBuffer' in r['Kernel_Name']]
ks.sort()
# last 8 persist launches: 2 warm-up + 5 timed (+ maybe)
p=[k for k in ks if 'persist' in k[2]]
t0=p[-int(__import__("os").environ.get("TRACE_LAST","7"))][0]
for s,e,n,q in ks:
    if s>=t0-2000000:
        print('%-14s q=%s start %9.3f ms  end %9.3f ms  dur %7.3f' % (n[:26].replace('void svo::',''),q,(s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6))
PY
tail -1 /tmp/tr.log | cut -c1-200
