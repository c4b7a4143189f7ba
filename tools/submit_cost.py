import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
from svo_raytracer_amd.cameras import CAMERAS
pool,_=scene.build_scene(512)
ctx=hiplib.HipContext(0); ctx.pool_upload(pool); ctx.resize(320,200); ctx.set_camera(CAMERAS["K1"]); ctx.set_pipeline(1); ctx.set_tuning(10,9)
ctx.set_params(2,0,0,0,2,0,1); ctx.ring_create(6,4,False)
for _ in range(12): ctx.ring_submit(2,4)
for b in range(6): ctx.ring_wait(b)
N=600
t=time.perf_counter()
for i in range(N):
    ctx.ring_submit(2+i,4)
dt=time.perf_counter()-t
for b in range(6): ctx.ring_wait(b)
print("host cost per svo_ring_submit (python ctypes included): %.1f us" % (dt/N*1e6))
import ctypes
L=hiplib.lib(); h=ctx._h; slot=ctypes.c_int()
t=time.perf_counter()
for i in range(N):
    L.svo_ring_submit(h, 2+i, 4, ctypes.byref(slot))
dt=time.perf_counter()-t
for b in range(6): ctx.ring_wait(b)
print("raw ctypes call: %.1f us" % (dt/N*1e6))
ctx.close()
