"""Timeline of the kernels of a short timed region (a driver's --steps 20 --warmup 5) from a rocprofv3 kernel trace:
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0
       --verify 0 --moving 0 --default-abi 0 --long-steps 0 --isolated 0
   python tools/burst_timeline.py <dir> [n launches, default 8]
Prints, for the last n launches of the persistent kernel and the table kernels in front of them: first wave, last wave (ms,
relative to the first of them), and how many of the launches are on the GPU over time."""
import csv, glob, os, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
pers = [r for r in rows if "persist_kernel" in r[2]]
pers = pers[-n:]
t0 = pers[0][0]
sel = [r for r in rows if r[0] >= t0 - 3_000_000 and ("persist_kernel" in r[2] or "rc_table" in r[2] or "zero_words" in r[2])]
t0 = min(r[0] for r in sel)
for s, e, k in sel:
    name = "persist" if "persist_kernel" in k else ("table" if "rc_table" in k else "zero")
    print("%-8s %9.3f -> %9.3f ms  (%7.3f)" % (name, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
ev = sorted([(r[0], 1) for r in pers] + [(r[1], -1) for r in pers])
cur = 0; last = ev[0][0]; acc = {}
for t, dlt in ev:
    acc[cur] = acc.get(cur, 0) + (t - last); last = t; cur += dlt
tot = sum(acc.values())
print("persistent launches on the GPU at once: " + ", ".join("%d: %.2f ms (%.0f %%)" % (k, v / 1e6, 100.0 * v / tot) for k, v in sorted(acc.items())))
print("first persistent wave to last: %.3f ms for %d launches" % ((pers[-1][1] - pers[0][0]) / 1e6, len(pers)))
