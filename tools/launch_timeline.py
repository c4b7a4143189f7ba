"""Where a single-frame launch of the persistent kernel spends its time (SVO_STAMPS build): when the band counters run dry
for the first / last wave, when the last wave ends.  SVO_HIP_LIB=.../libsvohip_stamps.so python tools/launch_timeline.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
from svo_raytracer_amd.cameras import CAMERAS
ctx = hiplib.HipContext(0)
h, m = scene.scene_maps(8192)
ctx.build_from_heightmap(h, m)
ctx.resize(1920, 1080); ctx.set_camera(CAMERAS["K1"]); ctx.set_hit_records(False); ctx.set_pipeline(1)
L = hiplib.lib()
L.svo_debug_heads.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
M = (1 << 64) - 1
for wpc in (8, 10, 14, 20):
    ctx.set_tuning(wpc, 9)
    for frame in (2, 3):
        ctx.set_params(frame, 0, 0, 0, 2, 0, 1)
        ms = ctx.time_frames(2, 1)
        buf = np.zeros(96, dtype=np.uint32)   # 16 x u64 of diagnostics + the 64-word histogram (svo_debug_heads: 384 bytes)
        L.svo_debug_heads(ctx._h, buf.ctypes.data)
        d = [int(x) for x in buf[:32].view(np.uint64)]
        nw = 256 * wpc
        begin = M - d[6]
        first_dry = M - d[7]
        us = lambda t: (t - begin) / 100.0
        mix = [(d[12 + i] & 0xffffffff, d[12 + i] >> 32) for i in range(4)]
        if frame == 2 and mix[0][0]:
            t = mix[0][0]
            print("  trips per frame %d (%.0f per wave), lane-iterations %d, rounds per wave %.1f" % (t, t / nw, mix[0][1], d[2] / nw))
            print("  per wave-trip: active lanes %.1f | descend section in %.1f %% of trips with %.1f lanes | advance %.1f %% with %.1f | pop %.1f %% with %.1f"
                  % (mix[0][1] / t, 100.0 * mix[1][0] / t, mix[1][1] / max(1, mix[1][0]), 100.0 * mix[2][0] / t, mix[2][1] / max(1, mix[2][0]),
                     100.0 * mix[3][0] / t, mix[3][1] / max(1, mix[3][0])))
        print("waves/CU %2d frame %d: %.3f ms by events | first wave out of work at %.0f us, last at %.0f us (mean %.0f), last wave ends %.0f us (mean end %.0f)"
              % (wpc, frame, ms[-1], us(first_dry), us(d[8]), d[11] / nw / 100.0, us(d[9]), d[10] / nw / 100.0))
