import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
from svo_raytracer_amd.cameras import CAMERAS
from oracle import oracle
big, _ = scene.build_scene(2048)
pool, _ = scene.build_scene(128)
cam = np.asarray(CAMERAS["K1"], dtype=np.float32)
w, h = 160, 96
ref = oracle.render(pool, w, h, cam, 3, 0)
def cmp(tag, res):
    print(tag, {k: int((res["hits"][k] != ref["hits"][k]).sum()) for k in ("pointer", "iter", "level")}, int((res["rgba"] != ref["rgba"]).any(axis=2).sum()))
for which in sys.argv[1:]:
    a = hiplib.HipContext(0)
    a.set_pipeline(int(which))
    a.render(big, 1920, 1080, cam, 2, 0)
    print("dirtied with pipeline", which, a.derived_info() if which == "1" else "")
    a.close()
    for p in (0, 1, 2):
        b = hiplib.HipContext(0)
        b.set_pipeline(p)
        cmp("fresh ctx pipeline %d" % p, b.render(pool, w, h, cam, 3, 0))
        cmp("  again", b.render(None, w, h, cam, 3, 0))
        b.close()
