"""Diagnosis (round 5): why does a 20-step region driven through the JNI-typed calls read lower than bench.py's own?"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
from svo_raytracer_amd.cameras import CAMERAS
from svo_raytracer_amd.framering import FrameRing
args = bench.parse(["--gpus", "1", "--steps", "20", "--warmup", "5", "--verify", "0"])
pool, _ = scene.build_scene(8192)
RAYS = 3306337.0
if os.environ.get("WHICH", "abi") == "abi":
    out = bench.run_default_abi(pool, 1920, 1080, CAMERAS["K1"], args, 6, 4, RAYS, 7)
    print("abi loop:", out["value"], out["value_repeats"], "long", out["value_long_run"])
else:
    ctx = hiplib.HipContext(0)
    if os.environ.get("POOL", "upload") == "build":
        hmap, mmap = scene.scene_maps(8192)
        ctx.build_from_heightmap(hmap, mmap)
    else:
        ctx.pool_upload(pool)
    ctx.resize(1920, 1080); ctx.set_camera(CAMERAS["K1"]); ctx.set_pipeline(1); ctx.derived_info()
    ring = FrameRing(ctx, 1920, 1080, nbuf=6, first_frame=2, batch=4, params=dict(render_mode=0, buffer_end=int(pool.size), bounces=2))
    def run(n):
        while n > 0:
            k = min(4, n); ring.step(k); n -= k
    run(1000); run(5)
    vals = []
    for _ in range(12):
        first = ring.dispatches % 6
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(20); torch.cuda.synchronize(); vals.append((first, round(RAYS * 20 / (time.perf_counter() - t0) / 1e6)))
    fresh = []
    for _ in range(8):      # the same burst on a ring made anew, with bench.py's own history: 4 + 1 warm-up frames, then 20
        ring = FrameRing(ctx, 1920, 1080, nbuf=6, first_frame=2, batch=4, params=dict(render_mode=0, buffer_end=int(pool.size), bounces=2))
        run(5)
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(20); torch.cuda.synchronize(); fresh.append(round(RAYS * 20 / (time.perf_counter() - t0) / 1e6))
    print("FrameRing loop, fresh ring per burst:", fresh)
    for what in ("count", "sleep", "persist"):
        vals = []
        for _ in range(6):      # what runs in front of the warm-up + burst: 20 counting passes (bench.py's history), 50 ms of nothing, 40 frames
            if what == "count":
                for fr in range(7, 27):
                    ctx.set_batch(1, 0); ctx.set_params(fr, 0, int(pool.size), 0, 2, 0, 1); ctx.count_frame()
            elif what == "sleep":
                torch.cuda.synchronize(); time.sleep(0.05)
            else:
                run(40)
            run(5)
            torch.cuda.synchronize(); t0 = time.perf_counter(); run(20); torch.cuda.synchronize(); vals.append(round(RAYS * 20 / (time.perf_counter() - t0) / 1e6))
        print("FrameRing loop, in front of the burst:", what, vals)
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(400); torch.cuda.synchronize()
    print("FrameRing loop:", vals, "long", round(RAYS * 400 / (time.perf_counter() - t0) / 1e6, 1))
