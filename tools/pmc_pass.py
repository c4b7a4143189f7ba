#!/usr/bin/env python3
"""Collect the per-launch PMC figures bench.py reports next to its roofline (GPU box only).

Runs the bench command under rocprofv3 in SEPARATE short passes (the guide: FETCH_SIZE and WRITE_SIZE do not fit
one pass; PMC runs carry no trace flags; whole batches only, so that every launch counted carries the same number of frames), averages each counter over the dispatches of the dominant kernel, and
writes gpurun_out/pmc_per_launch.json keyed like bench.py keys it, stamped with the hash of the kernel sources:
bench.py only reports figures whose hash matches the sources it runs (copy the file to profiles/ to commit it).

usage: python3 tools/pmc_pass.py [--kernel 'persist_kernel<0, svo::DescWalk, false, true>'] [--tag r05] -- <bench.py args>
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PASSES = [
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE"]),
    ("valu", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
              "SQ_WAIT_INST_ANY", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE"]),
]


def main():
    argv = sys.argv[1:]
    kernel, tag = "persist_kernel<0, svo::DescWalk, false, true>", "r05"
    while argv and argv[0] != "--":
        if argv[0] == "--kernel":
            kernel = argv[1]
        elif argv[0] == "--tag":
            tag = argv[1]
        argv = argv[2:]
    bench_args = argv[1:] if argv else []
    import bench
    a = bench.parse(bench_args)
    nbuf = min(8, max(1, a.inflight))
    batch = a.batch if a.batch > 0 else bench.default_batch(a, 1)
    key = bench.pmc_key(a, a.width, a.height, nbuf, batch)
    out_root = os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (tag, key))
    os.makedirs(out_root, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    means, ndisp = {}, {}
    for name, counters in PASSES:
        d = os.path.join(out_root, name)
        subprocess.call(["rm", "-rf", d])
        cmd = ["timeout", "-s", "KILL", "300", "rocprofv3", "--pmc"] + counters + \
              ["--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"),
               "--steps", str(8 * batch), "--warmup", str(batch), "--cpu-seconds", "0", "--verify", "0", "--isolated", "0", "--moving", "0", "--default-abi", "0", "--long-steps", "0"] + bench_args
        with open(os.path.join(out_root, name + ".log"), "w") as lf:
            rc = subprocess.call(cmd, cwd="/tmp", env=env, stdout=lf, stderr=subprocess.STDOUT)
        print("pass %s rc %d" % (name, rc), flush=True)
        agg, cnt = collections.defaultdict(float), collections.defaultdict(int)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel in r["Kernel_Name"]:
                    agg[r["Counter_Name"]] += float(r["Counter_Value"])
                    cnt[r["Counter_Name"]] += 1
        for c in agg:
            means[c] = agg[c] / cnt[c]
            ndisp[c] = cnt[c]
    need = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU")
    missing = [c for c in need if c not in means]
    if missing:
        print("missing counters:", missing)
        sys.exit(1)
    entry = {
        "src_hash": bench.source_hash(), "kernel": kernel, "dispatches_averaged": ndisp,
        "fetch_size_kb": means["FETCH_SIZE"], "write_size_kb": means["WRITE_SIZE"],
        "sq_insts_valu": means["SQ_INSTS_VALU"], "sq_active_inst_valu": means["SQ_ACTIVE_INST_VALU"],
        "sq_thread_cycles_valu": means["SQ_THREAD_CYCLES_VALU"],
        "other": {k: v for k, v in means.items() if k not in need},
        "how": "rocprofv3 --pmc, separate passes of `bench.py --steps %d --warmup %d --verify 0 --isolated 0 " % (8 * batch, batch) + " ".join(bench_args) +
               "`, per-dispatch mean over the kernel's launches; traffic = FETCH_SIZE x 2 (gfx950: 128-B requests "
               "tallied at 64 B, profiles/r01_fetch_size_calibration.txt) + WRITE_SIZE, both in KB",
    }
    path = os.path.join(ROOT, "gpurun_out", "pmc_per_launch.json")
    allj = {}
    if os.path.exists(path):
        try:
            allj = json.load(open(path))
        except Exception:
            allj = {}
    allj[key] = entry
    json.dump(allj, open(path, "w"), indent=1)
    print(json.dumps({key: entry}, indent=1))


if __name__ == "__main__":
    main()
