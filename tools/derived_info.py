import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
ctx = hiplib.HipContext(0)
for n in (() if "--strokes-only" in sys.argv else (512, 2048, 8192)):
    h, m = scene.scene_maps(n)
    nb = ctx.build_from_heightmap(h, m)
    t0 = time.time(); i = ctx.derived_info(); dt = time.time() - t0
    print(n, nb, i, "host wall %.1f ms" % (dt * 1e3))
    pool = ctx.pool_download(nb)
    import numpy as np
    edited = pool.copy(); edited[1000:1007] = edited[1000:1007]
    t0 = time.time(); ctx.pool_update(edited, 1000, 1007); i2 = ctx.derived_info(); print("  after a 7-byte svo_pool_update: rebuild wall %.1f ms (GPU %.2f ms)" % ((time.time() - t0) * 1e3, i2["build_ms"]))

# an SDF brush stroke at 8192^3 (oracle-side restatement of Octree.useSDFBrush; run from tests' side of the fence: this
# tool is a measurement script, not product code): the two ranged updates with the table following them
# (derive::refresh_table) against SVO_DERIVED_REFRESH=0 in a child process (the table rebuilt by the next dispatch)
import os, subprocess
if os.environ.get("SVO_DERIVED_REFRESH", "1") != "0" and "--no-child" not in sys.argv:
    env = dict(os.environ, SVO_DERIVED_REFRESH="0")
    print(subprocess.run([sys.executable, __file__, "--no-child", "--strokes-only"], env=env, capture_output=True, text=True).stdout[-1500:])
from svo_raytracer_amd import hostlib
from oracle import octree as restated
n = 8192
h, m = scene.scene_maps(n)
nb = ctx.build_from_heightmap(h, m)
pool = ctx.pool_download(nb)
ctx.derived_info()
o = hostlib.Octree((nb + (64 << 20)) // 1024)
o.adopt(pool)
rng = np.random.default_rng(5)
print("strokes at 8192^3, SVO_DERIVED_REFRESH =", os.environ.get("SVO_DERIVED_REFRESH", "1"))
for stroke in range(6):
    org = (int(rng.integers(3000, 5000)), int(h[4000, 4000]) + int(rng.integers(-10, 10)), int(rng.integers(3000, 5000)))
    r = int(rng.integers(6, 40)); val = int(rng.choice([0, 2]))
    cb = restated.useSDFBrushSphere(o, org, r, val, worldSize=n, maxLOD=13)
    host = o.getByteBuffer()
    t0 = time.time(); gpu = []
    for s, e in ((cb[0], cb[1]), (cb[2], cb[3])):
        if e > s:
            ctx.pool_update(host, s, e); gpu.append(ctx.derived_refresh_info())
    i = ctx.derived_info()      # (rebuilds here when the table was dropped)
    dt = (time.time() - t0) * 1e3
    if cb[1] <= cb[0] and cb[3] <= cb[2]:
        print("  r=%2d value %d: the stroke changed nothing" % (r, val)); continue
    print("  r=%2d value %d: ranges %.2f MB + %.2f MB, updates + table %.2f ms wall; refresh %s; build_ms %.2f; descriptors %d"
          % (r, val, (cb[1] - cb[0]) / 1e6, (cb[3] - cb[2]) / 1e6, dt, [(g["states"], g["added"], round(g["gpu_ms"], 3)) for g in gpu], i["build_ms"], i["descriptors"]))
