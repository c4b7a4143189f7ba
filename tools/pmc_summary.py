#!/usr/bin/env python3
"""Summarise rocprofv3 csv outputs (kernel stats + PMC counters per kernel) into one text file."""
import csv, glob, os, sys, collections
root = sys.argv[1]
out = []
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
    out.append("== " + os.path.relpath(f, root))
    out += [l.rstrip() for l in open(f)][:12]
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    out.append("== " + os.path.relpath(f, root) + "  (per-dispatch mean)")
    for k in agg:
        for c, v in sorted(agg[k].items()):
            n = cnt[(k, c)]
            out.append("%-60s %-24s n=%-5d mean=%.6g" % (k, c, n, v / n))
print("\n".join(out))
