#!/usr/bin/env python3
"""The measurement matrix of round 6: SURVEY 8(d)'s three cameras x three scenes, each cell the default
bench configuration (8192^3, 1920x1080, renderMode 0, 6 submissions in flight x 4 frames, 400 verified steps) plus the
lanes-per-trip figures of an -DSVO_STAMPS=1 build.  GPU box only, from the repository root:

    python tools/matrix.py [--out gpurun_out/matrix] [--cells all|t1a8_K1,...] [--stamps 1] [--sweep CELL]

writes <out>/<cell>.json (the bench line), <out>/stamps.json and <out>/matrix.md (copy to profiles/round6_matrix.md).
--sweep CELL: --waves 8/10/12 x --thresh 8/9/10 on that cell, 400 steps each, twice -> <out>/sweep_<cell>.md."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = {"t1a8": ("terrain", 1, 8), "t2a18": ("terrain", 2, 18), "c1a8d64": ("caves", 1, 8), "d1a8x24": ("dust", 1, 8)}
SCENE_TEXT = {"t1a8": "terrain, seed 1, amplitude 8/16 (rounds 1-5)", "t2a18": "terrain, seed 2, amplitude 18/16",
              "c1a8d64": "caves: terrain seed 1 + hashed balls (dens 64/256)",
              "d1a8x24": "dust: terrain seed 1 + floating particles (24/256 of the air cells)"}
CAMS = ("K0", "K1", "K2")
BENCH = ["--steps", "400", "--warmup", "24", "--long-steps", "0", "--moving", "0", "--default-abi", "0", "--by-camera", "0", "--ref-loop", "1", "--cpu-seconds", "0"]


def bench_args(cell):
    sk, cam = cell.split("_")
    family, seed, amp = SCENES[sk]
    return ["--scene", family, "--seed", str(seed), "--amp", str(amp), "--camera", cam]


def run_bench(cell, extra, env):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + bench_args(cell) + BENCH + extra
    r = subprocess.run(cmd, capture_output=True, env=env, cwd=ROOT)
    line = None
    for ln in r.stdout.decode().splitlines():
        if ln.startswith("{"):
            line = json.loads(ln)
    if line is None:
        sys.stderr.write("%s: no line (rc %d): %s\n" % (cell, r.returncode, r.stderr.decode()[-400:]))
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "matrix"))
    ap.add_argument("--cells", default="all")
    ap.add_argument("--stamps", type=int, default=1)
    ap.add_argument("--sweep", default=None)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    cache = "/tmp/svo_scene_cache"
    os.makedirs(cache, exist_ok=True)
    env = dict(os.environ, SVO_SCENE_CACHE=cache)
    if a.sweep:
        rows = []
        for rep in range(2):
            for w in (8, 10, 12):
                for t in (8, 9, 10):
                    j = run_bench(a.sweep, ["--waves", str(w), "--thresh", str(t), "--verify", "0" if rep else "1", "--isolated", "0", "--ref-loop", "0"], env)
                    rows.append((w, t, rep, j["value"] if j else None))
                    print(rows[-1], flush=True)
        with open(os.path.join(a.out, "sweep_%s.md" % a.sweep), "w") as f:
            f.write("| waves per CU | threshold /16 | Mrays/s (two passes) | mean |\n|---|---|---|---|\n")
            for w in (8, 10, 12):
                for t in (8, 9, 10):
                    v = [r[3] for r in rows if r[0] == w and r[1] == t and r[3]]
                    f.write("| %d | %d | %s | %.0f |\n" % (w, t, " / ".join("%.0f" % x for x in v), sum(v) / max(len(v), 1)))
        return
    cells = ([sk + "_" + c for sk in SCENES for c in CAMS] + ["c1a8d64_CAVE"]) if a.cells == "all" else a.cells.split(",")
    lines = {}
    for cell in cells:
        j = run_bench(cell, [], env)
        if j:
            lines[cell] = j
            json.dump(j, open(os.path.join(a.out, cell + ".json"), "w"))
            print(cell, j["value"], j["verified"], flush=True)
    stamps = {}
    lib = os.path.join(ROOT, "svo-raytracer_amd", "csrc", "libsvohip_stamps.so")
    if a.stamps and os.path.exists(lib):
        for cell in cells:
            sk, cam = cell.split("_")
            family, seed, amp = SCENES[sk]
            e = dict(env, STAMPS_JSON="1", STAMPS_WAVES="10", STAMPS_BATCH="4", STAMPS_SCENE=family, STAMPS_SEED=str(seed), STAMPS_AMP=str(amp),
                     STAMPS_CAMERA=cam, SVO_HIP_LIB=lib)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stamps.py")], capture_output=True, env=e, cwd=ROOT)
            open(os.path.join(a.out, "stamps_%s.txt" % cell), "w").write(r.stdout.decode() + r.stderr.decode()[-2000:])
        try:
            allj = json.load(open(os.path.join(ROOT, "gpurun_out", "stamps_per_launch.json")))
            for cell in cells:
                sk, cam = cell.split("_")
                family, seed, amp = SCENES[sk]
                stamps[cell] = allj.get("%s_s%d_a%d_%s" % (family, seed, amp, cam))
        except Exception as ex:     # noqa: BLE001
            sys.stderr.write("stamps: %s\n" % ex)
        json.dump(stamps, open(os.path.join(a.out, "stamps.json"), "w"), indent=1)
    with open(os.path.join(a.out, "matrix.md"), "w") as f:
        f.write("| scene | camera | Mrays/s | ms/frame | rays/frame | iterations/ray | B_alg/ray | frac of 8 TB/s | one frame at a time, Mrays/s | "
                "lanes traversing per trip (of 64) | descend / advance / pop: share of trips, lanes | pool, GB | verified |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for cell in cells:
            j = lines.get(cell)
            if not j:
                f.write("| %s | %s | failed |\n" % tuple(cell.split("_")))
                continue
            sk, cam = cell.split("_")
            c, s = j["config"], stamps.get(cell) or {}
            sec = s.get("sections") or {}
            sect = " / ".join("%s %%, %s" % (sec[n]["share_of_trips_pct"], sec[n]["mean_lanes"]) for n in ("descend", "advance", "pop")) if sec else "-"
            nbytes = int(c["workload"].split(" bytes")[0].split(", ")[-1])
            f.write("| %s | %s | %.0f | %.4f | %d | %.1f | %.1f | %.3f | %s | %s | %s | %.2f | %s |\n" % (
                SCENE_TEXT[sk], cam, j["value"], j["ms_per_step"], c["rays_per_frame"], c["iterations_per_ray"], c["alg_bytes_per_ray"],
                j["roofline"]["frac"], "%.0f" % j["value_one_frame_at_a_time"] if j.get("value_one_frame_at_a_time") else "-",
                s.get("mean_lanes_traversing", "-"), sect, nbytes / 1e9, j["verified"]))
    print(open(os.path.join(a.out, "matrix.md")).read())


if __name__ == "__main__":
    main()
