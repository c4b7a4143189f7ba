import sys; sys.path.insert(0, "/root/repo")
import time
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
ctx = hiplib.HipContext(0)
for n in (512, 2048, 8192):
    h, m = scene.scene_maps(n)
    nb = ctx.build_from_heightmap(h, m)
    t0 = time.time(); i = ctx.derived_info(); dt = time.time() - t0
    print(n, nb, i, "host wall %.1f ms" % (dt * 1e3))
    pool = ctx.pool_download(nb)
    import numpy as np
    edited = pool.copy(); edited[1000:1007] = edited[1000:1007]
    t0 = time.time(); ctx.pool_update(edited, 1000, 1007); i2 = ctx.derived_info(); print("  after a 7-byte svo_pool_update: rebuild wall %.1f ms (GPU %.2f ms)" % ((time.time() - t0) * 1e3, i2["build_ms"]))
