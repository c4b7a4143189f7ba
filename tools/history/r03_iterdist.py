"""Distribution of primary-ray iteration counts over the bench frame (8192^3, 1080p, K1) and over its 8x8 tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
from svo_raytracer_amd.cameras import CAMERAS
ctx = hiplib.HipContext(0)
h, m = scene.scene_maps(8192)
ctx.build_from_heightmap(h, m)
ctx.set_pipeline(1)
for cam in ("K1", "K0", "K2"):
    r = ctx.render(None, 1920, 1080, CAMERAS[cam], 2, 0)
    it = r["hits"]["iter"].astype(np.int64)
    print(cam, "primary iterations: mean %.1f p50 %d p90 %d p99 %d p99.9 %d max %d" % (it.mean(), *np.percentile(it, [50, 90, 99, 99.9]).astype(int), it.max()))
    t = it[:1080 // 8 * 8].reshape(135, 8, 240, 8)
    tmax = t.max(axis=(1, 3)); tsum = t.sum(axis=(1, 3))
    print("  tiles: max-of-tile p50 %d p90 %d p99 %d max %d; tile sums p50 %d p90 %d p99 %d max %d" % (
        *np.percentile(tmax, [50, 90, 99]).astype(int), tmax.max(), *np.percentile(tsum, [50, 90, 99]).astype(int), tsum.max()))
    rows = tmax.max(axis=1)
    print("  per tile row max:", " ".join(str(int(x)) for x in rows[::6]))
    rows = tsum.mean(axis=1)
    print("  per tile row mean sum:", " ".join(str(int(x)) for x in rows[::6]))
