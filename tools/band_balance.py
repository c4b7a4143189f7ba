import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
from svo_raytracer_amd.cameras import CAMERAS
from svo_raytracer_amd.tiles import band_rows
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
W, H = 1920, 1080
pool, _ = scene.build_scene(n)
ctx = hiplib.HipContext(0)
ctx.pool_upload(pool); ctx.resize(W, H); ctx.set_hit_records(False)
for camname in ("K0", "K1", "K2"):
    ctx.set_camera(CAMERAS[camname]); ctx.set_params(2, 0, 0, 0, 2, 0, 1)
    for pipe in (1,):
        ctx.set_pipeline(pipe)
        ctx.set_rows(0, H)
        full = float(np.median(ctx.time_frames(3, 20)))
        for world in (2, 4, 8):
            ts = []
            for r in range(world):
                y0, y1, _ = band_rows(H, world, r)
                ctx.set_rows(y0, y1)
                ts.append(float(np.median(ctx.time_frames(2, 10))))
            print(camname, "pipe", pipe, "full %.3f ms" % full, "world", world, "bands", " ".join("%.3f" % t for t in ts),
                  "max %.3f -> strong-scaling bound %.2fx of %d" % (max(ts), full / max(ts), world))
