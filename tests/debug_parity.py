import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
from svo_raytracer_amd.cameras import CAMERAS
from oracle import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W, H = 1920, 1080
pool, _ = scene.build_scene(n)
ctx = hiplib.HipContext(0)
a = ctx.render(pool, W, H, CAMERAS["K1"], 2, mode)
ref = oracle.render(pool, W, H, CAMERAS["K1"], 2, mode)
for k in ("pointer", "value", "raw_normal", "level", "iter"):
    bad = a["hits"][k] != ref["hits"][k]
    print(k, int(bad.sum()))
bad = a["hits"]["pointer"] != ref["hits"]["pointer"]
print("rgba", int((a["rgba"] != ref["rgba"]).any(axis=2).sum()), "depth", int((a["depth"].view(np.uint32) != ref["depth"].view(np.uint32)).sum()))
ys, xs = np.nonzero(bad | (a["hits"]["iter"] != ref["hits"]["iter"]))
for y, x in list(zip(ys, xs))[:12]:
    print(x, y, "gpu", a["hits"][y, x], "ref", ref["hits"][y, x])
print("stats", ref["stats"])
