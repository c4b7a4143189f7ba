import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
from svo_raytracer_amd.cameras import CAMERAS
# the cell: STAMPS_SCENE = terrain | caves | dust, STAMPS_SEED, STAMPS_AMP, STAMPS_CAMERA = K0 | K1 | K2 (default: the bench line's)
SCENE, SEED, AMP = os.environ.get("STAMPS_SCENE", "terrain"), int(os.environ.get("STAMPS_SEED", "1")), int(os.environ.get("STAMPS_AMP", "8"))
CAMERA = os.environ.get("STAMPS_CAMERA", "K1")
CELL = "%s_s%d_a%d_%s" % (SCENE, SEED, AMP, CAMERA)
_cache = os.environ.get("SVO_SCENE_CACHE") and os.path.join(os.environ["SVO_SCENE_CACHE"], "%s_8192_s%d_a%d_d%d.npy" % (
    SCENE, SEED, AMP, scene.DUST_DENS if SCENE == "dust" else scene.CAVES_DENS))
ctx = hiplib.HipContext(0)
if SCENE == "terrain":      # on the GPU from its two maps, as bench.py does
    ctx.build_from_heightmap(*scene.scene_maps(8192, SEED, AMP))
elif _cache and os.path.exists(_cache):
    ctx.pool_upload(np.load(_cache))
else:
    ctx.pool_upload(scene.build(SCENE, 8192, SEED, AMP)[0])
if CAMERA == "CAVE":
    from svo_raytracer_amd.cameras import cave_camera
    CAMERAS = dict(CAMERAS, CAVE=cave_camera(8192, SEED, AMP))
ctx.resize(1920, 1080); ctx.set_camera(CAMERAS[CAMERA]); ctx.set_hit_records(False); ctx.set_pipeline(1)
L = hiplib.lib()
L.svo_debug_heads.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
import os, json
OUT = {}
BATCH = int(os.environ.get("STAMPS_BATCH", "1"))     # frames per launch (through a ring of one slot when > 1)
for wpc in [int(v) for v in os.environ.get("STAMPS_WAVES", "10,20").split(",")]:
    t = int(os.environ.get("STAMPS_THRESH", "0"))    # 0 = the kernel's own default
    ctx.set_tuning(wpc, t)
    ctx.set_params(2, 0, 0, 0, 2, 0, 1)
    if BATCH > 1:
        ctx.ring_create(1, BATCH)
        for k in range(2):
            sl = ctx.ring_submit(2 + k * BATCH, BATCH)
            ctx.ring_wait(sl)
        ms = [ctx.ring_query(sl)["gpu_ms"] / BATCH]
    else:
        ms = ctx.time_frames(2, 1)   # 3 launches -> ring slots k, k+1, k+2; read all, take the last launch's set
    buf = np.zeros(176, dtype=np.uint32)
    L.svo_debug_heads(ctx._h, buf.ctypes.data)
    d = buf[0:12].view(np.uint64)
    nw = 256 * wpc
    mixw = buf[24:32].view(np.uint64)   # lanes << 32 | trips: whole trip / descend / advance / pop
    trips = int(mixw[0] & 0xffffffff)
    sect = {n: (int(mixw[i] & 0xffffffff), int(mixw[i] >> 32)) for i, n in enumerate(("trip", "descend", "advance", "pop"))}
    hist = buf[32:96].astype(np.float64)   # [L] = trips with exactly L lanes traversing (64 lanes: the rest)
    h64 = max(trips - hist.sum(), 0.0)
    full = np.concatenate([hist, [h64]])
    pop_hist = buf[96:160].astype(np.float64)   # [L] = trips whose POP section ran for exactly L lanes (L >= 1)
    pop = None
    if trips and pop_hist.sum() > 0:
        pfull = np.concatenate([[max(trips - pop_hist.sum(), 0.0)], pop_hist[1:]])   # [0] = trips without a POP lane
        pc = np.cumsum(pfull) / trips
        pop = {"trips_without_pop_pct": round(100 * pfull[0] / trips, 1), "trips_with_1_to_5_pop_lanes_pct": round(100 * (pc[5] - pc[0]), 1),
               "trips_with_6_to_15_pct": round(100 * (pc[15] - pc[5]), 1), "trips_with_16_or_more_pct": round(100 * (1 - pc[15]), 1),
               "mean_pop_lanes_when_it_runs": round(float((pfull[1:] * np.arange(1, 64)).sum() / max(pfull[1:].sum(), 1)), 1)}
        print("  POP section: %s" % pop)
    parts = buf[160:176].view(np.uint64).astype(np.float64)   # cycles of all rounds by part, then lanes shaded / set up / refilled
    nr = max(float(d[2]), 1.0)
    rparts = {"cast_result_cycles": round(parts[0] / nr), "shade_cycles": round(parts[1] / nr), "store_cycles": round(parts[2] / nr),
              "refill_cycles": round(parts[3] / nr), "ray_setup_cycles": round(parts[4] / nr), "round_cycles": round(float(d[0]) / nr),
              "lanes_shaded": round(parts[5] / nr, 1), "lanes_set_up": round(parts[6] / nr, 1), "lanes_refilled": round(parts[7] / nr, 1)}
    print("  a round by part (cycles of the wave, per round):", rparts)
    if trips:
        lanes = np.arange(65)
        mean = (full * lanes).sum() / full.sum()
        cum = np.cumsum(full) / full.sum()
        print("  lanes traversing per trip: mean %.1f; trips with < 24 / 32 / 40 / 48 / 56 lanes: %.1f / %.1f / %.1f / %.1f / %.1f %%; idle lane-trips %.1f %% of 64 x trips" % (
            mean, 100 * cum[23], 100 * cum[31], 100 * cum[39], 100 * cum[47], 100 * cum[55], 100 * (1 - mean / 64)))
        OUT["waves%d_batch%d" % (wpc, BATCH)] = {
            "mean_lanes_traversing": round(float(mean), 2), "idle_lane_trips_pct": round(100 * (1 - mean / 64), 2),
            "trips_below_40_lanes_pct": round(100 * float(cum[39]), 2), "trips_with_all_64_pct": round(100 * float(full[64] / full.sum()), 2),
            "trips_per_wave": round(trips / nw, 1), "rounds_per_wave": round(float(d[2]) / nw, 1),
            "sections": {n: {"share_of_trips_pct": round(100.0 * t / max(trips, 1), 1), "mean_lanes": round(l / max(t, 1), 1)} for n, (t, l) in sect.items()},
            "ms_per_frame": round(float(ms[-1]), 4), "cell": CELL, "pop_section": pop, "round_parts": rparts,
            "what": "SVO_STAMPS build of the same sources: per trip of the assembly loop, lanes traversing; %d persistent waves per CU, "
                    "%d frame(s) per launch, one launch at a time, 8192^3 %s seed %d amp %d / 1920x1080 / mode 0 / %s" % (wpc, BATCH, SCENE, SEED, AMP, CAMERA)}
        print("  histogram by 8 lanes:", " ".join("%.1f" % (100 * full[i:i + 8].sum() / full.sum()) for i in range(0, 64, 8)), "| 64: %.1f" % (100 * full[64] / full.sum()))
    if d[3] == 0:   # assembly loop: trips are not counted inside the asm block
        print("waves/cu", wpc, "thresh", t, "batch", BATCH, "ms/frame %.3f" % ms[-1], "rounds/wave %.1f trips/wave %.0f  cyc/round: round %.0f (shade part %.0f) + traversal %.0f; cyc/trip %.0f; round share %.1f %%" % (
            d[2] / nw, trips / nw, float(d[0]) / max(d[2], 1), float(d[4]) / max(d[2], 1), float(d[1]) / max(d[2], 1), float(d[1]) / max(trips, 1),
            100.0 * float(d[0]) / max(float(d[0]) + float(d[1]), 1)))
        continue
    print("waves/cu", wpc, "thresh", t, "ms %.3f" % ms[-1], "rounds/wave %.1f trips/wave %.1f  cyc/round %.0f (shade %.0f, refill+init %.0f)  cyc/trip %.0f (of which load issue->data %.0f)" % (
        d[2] / nw, d[3] / nw, d[0] / max(d[2], 1), d[4] / max(d[2], 1), (d[0] - d[4]) / max(d[2], 1), d[1] / max(d[3], 1), d[5] / max(d[3], 1)))

if os.environ.get("STAMPS_JSON"):
    import bench
    key = "waves10_batch4" if "waves10_batch4" in OUT else sorted(OUT)[0]
    e = dict(OUT[key], src_hash=bench.source_hash(), all=OUT)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "stamps_per_launch.json")
    allj = {}
    if os.path.exists(path):
        try:
            allj = json.load(open(path))
        except Exception:
            allj = {}
    allj[CELL] = e
    # ... and under the key bench.py looks its own configuration up by (the default launch shape: 6 submissions in flight x 4 frames)
    bargs = ["--scene", SCENE, "--seed", str(SEED), "--amp", str(AMP), "--camera", CAMERA]
    _lib = os.environ.pop("SVO_HIP_LIB", None)      # (the stamps build stands for the product library's kernels: same sources, hash-gated)
    allj[bench.pmc_key(bench.parse(bargs), 1920, 1080, 6, BATCH)] = e
    if _lib is not None:
        os.environ["SVO_HIP_LIB"] = _lib
    json.dump(allj, open(path, "w"), indent=1)
    print("wrote", path, CELL)
