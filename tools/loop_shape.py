#!/usr/bin/env python3
"""The reference's loop (bench.py::run_reference_loop: nSetCamera, nSetParams, nDispatchAsync, nReadPixel at the crosshair, wall clock)
by launch shape: persistent waves per CU of a dispatch (svo_set_tuning; 0 = the library's own choice: 12 while svo_dispatch_async alternates its two
image sets, as many as fit -- 24 -- otherwise).  GPU box only:  python tools/loop_shape.py [waves ...]   -> gpurun_out/loop_shape.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = [int(v) for v in sys.argv[1:]] or [0, 24, 20, 16, 14, 12, 10]
sets = [int(v) for v in os.environ.get("LOOP_SETS", "0").split(",")]     # LOOP_SETS=2,3,4: also by number of image sets
rows = []
for rep in range(2):
  for ns in sets:
    for w in shapes:
        env = dict(os.environ)
        if w:
            env["SVO_LOOP_WAVES"] = str(w)
        if ns:
            env["SVO_LOOP_SETS"] = str(ns)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "8", "--long-steps", "0", "--moving", "0",
                            "--default-abi", "0", "--by-camera", "0", "--ref-loop", "1", "--cpu-seconds", "0", "--isolated", "0"],
                           capture_output=True, env=env, cwd=ROOT)
        line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
        j = json.loads(line[-1])["reference_loop"] if line else {}
        rows.append((w, rep, j.get("static", {}).get("value"), j.get("moving", {}).get("value"), j.get("without_overlap", {}).get("value"), ns))
        print(rows[-1], flush=True)
with open(os.path.join(ROOT, "gpurun_out", "loop_shape.txt"), "w") as f:
    f.write("image sets (0 = the library's) | waves per CU (0 = the library's) | static | moving | without overlap   (Mrays/s, two passes)\n")
    for ns in sets:
        for w in shapes:
            v = [r for r in rows if r[0] == w and r[5] == ns]
            f.write("%d | %3d | %s | %s | %s\n" % (ns, w, " / ".join(str(r[2]) for r in v), " / ".join(str(r[3]) for r in v), " / ".join(str(r[4]) for r in v)))
print(open(os.path.join(ROOT, "gpurun_out", "loop_shape.txt")).read())
