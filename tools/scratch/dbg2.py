import sys, os, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
from svo_raytracer_amd.cameras import CAMERAS
from oracle import oracle
import test_config3 as t3
pool8192 = scene.build_scene(8192)[0]
pool2048 = scene.build_scene(2048)[0]
which = sys.argv[1:]
for name in which:
    fn = getattr(t3, name)
    print("running", name, flush=True)
    if "pipeline" in fn.__code__.co_varnames[:fn.__code__.co_argcount]:
        for p in (0, 1, 2):
            fn(pool2048 if "config2" in name else pool8192, p)
    else:
        fn(pool8192)
L = hiplib.lib()
vp, jint, jlong, jfloat = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
P = "Java_src_engine_HipRenderer_"
def fn(name, res, *args):
    f = getattr(L, P + name); f.restype = res; f.argtypes = [vp, vp] + list(args)
    return lambda *a: f(None, None, *a)
nCreate = fn("nCreate", jlong, jint); nDestroy = fn("nDestroy", jint, jlong)
nPoolUpload = fn("nPoolUpload", jint, jlong, jlong, jlong)
nSetCamera = fn("nSetCamera", jint, jlong, *([jfloat] * 15))
nSetParams = fn("nSetParams", jint, jlong, *([jint] * 7))
nResize = fn("nResize", jint, jlong, jint, jint)
nDispatch = fn("nDispatch", jint, jlong)
nReadHits = fn("nReadHits", jint, jlong, jlong)
pool, _ = scene.build_scene(128)
cam = np.asarray(CAMERAS["K1"], dtype=np.float32)
w, h = 160, 96
ref = oracle.render(pool, w, h, cam, 3, 0)
for rep in range(3):
    j = nCreate(0)
    print("dispatch without pool:", nDispatch(j))
    assert nPoolUpload(j, pool.ctypes.data, pool.size) == 0
    assert nSetCamera(j, *[float(v) for v in cam]) == 0
    assert nSetParams(j, 3, 0, int(pool.size), 0, 2, 0, 1) == 0
    assert nResize(j, w, h) == 0
    for d in range(2):
        assert nDispatch(j) == 0
        hits = np.zeros((h, w), dtype=hiplib.HIT_DTYPE)
        assert nReadHits(j, hits.ctypes.data) == 0
        bad = hits["iter"] != ref["hits"]["iter"]
        print("rep", rep, "dispatch", d, "iter mismatches", int(bad.sum()), "pointer", int((hits["pointer"] != ref["hits"]["pointer"]).sum()))
        if bad.any():
            ys, xs = np.nonzero(bad)
            print("  rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "sample got/want", hits["iter"][bad][:8], ref["hits"]["iter"][bad][:8])
    nDestroy(j)
