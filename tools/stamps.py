import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
from svo_raytracer_amd.cameras import CAMERAS
pool, _ = scene.build_scene(8192)
ctx = hiplib.HipContext(0)
ctx.pool_upload(pool); ctx.resize(1920, 1080); ctx.set_camera(CAMERAS["K1"]); ctx.set_hit_records(False); ctx.set_pipeline(1)
L = hiplib.lib()
L.svo_debug_heads.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
for wpc in (10, 20):
    t = 9
    ctx.set_tuning(wpc, t)
    ctx.set_params(2, 0, 0, 0, 2, 0, 1)
    ms = ctx.time_frames(2, 1)   # 3 launches -> ring slots k, k+1, k+2; read all, take the last launch's set
    buf = np.zeros(96, dtype=np.uint32)
    L.svo_debug_heads(ctx._h, buf.ctypes.data)
    d = buf[0:12].view(np.uint64)
    nw = 256 * wpc
    mixw = buf[24:32].view(np.uint64)   # lanes << 32 | trips: whole trip / descend / advance / pop
    trips = int(mixw[0] & 0xffffffff)
    hist = buf[32:96].astype(np.float64)   # [L] = trips with exactly L lanes traversing (64 lanes: the rest)
    h64 = max(trips - hist.sum(), 0.0)
    full = np.concatenate([hist, [h64]])
    if trips:
        lanes = np.arange(65)
        mean = (full * lanes).sum() / full.sum()
        cum = np.cumsum(full) / full.sum()
        print("  lanes traversing per trip: mean %.1f; trips with < 24 / 32 / 40 / 48 / 56 lanes: %.1f / %.1f / %.1f / %.1f / %.1f %%; idle lane-trips %.1f %% of 64 x trips" % (
            mean, 100 * cum[23], 100 * cum[31], 100 * cum[39], 100 * cum[47], 100 * cum[55], 100 * (1 - mean / 64)))
        print("  histogram by 8 lanes:", " ".join("%.1f" % (100 * full[i:i + 8].sum() / full.sum()) for i in range(0, 64, 8)), "| 64: %.1f" % (100 * full[64] / full.sum()))
    if d[3] == 0:   # assembly loop: trips are not counted inside the asm block
        print("waves/cu", wpc, "thresh", t, "ms %.3f" % ms[-1], "rounds/wave %.1f  cyc/round: shade %.0f + refill+init %.0f + traversal %.0f" % (
            d[2] / nw, d[4] / max(d[2], 1), (d[0] - d[4]) / max(d[2], 1), d[1] / max(d[2], 1)))
        continue
    print("waves/cu", wpc, "thresh", t, "ms %.3f" % ms[-1], "rounds/wave %.1f trips/wave %.1f  cyc/round %.0f (shade %.0f, refill+init %.0f)  cyc/trip %.0f (of which load issue->data %.0f)" % (
        d[2] / nw, d[3] / nw, d[0] / max(d[2], 1), d[4] / max(d[2], 1), (d[0] - d[4]) / max(d[2], 1), d[1] / max(d[3], 1), d[5] / max(d[3], 1)))
