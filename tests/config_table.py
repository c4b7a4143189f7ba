#!/usr/bin/env python3
"""Fill the result table of BASELINE.md section 4 for the five BASELINE.json configs on ONE GPU
(one frame at a time, HIP-event kernel time; the headline bench keeps 3 frames in flight)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
from svo_raytracer_amd.cameras import CAMERAS
from oracle import oracle

CONFIGS = [
    # name, N, W, H, mode, bounces, mirror, spp, camera, oracle subsample step
    ("C1 512^3 256x256 primary (mode 1)", 512, 256, 256, 1, 2, 0, 1, "K0", 1),
    ("C2 2048^3 1920x1080 primary (mode 1)", 2048, 1920, 1080, 1, 2, 0, 1, "K1", 8),
    ("C3 8192^3 1920x1080 primary + 1 bounce", 8192, 1920, 1080, 0, 2, 0, 1, "K1", 16),
    ("C4 8192^3 3840x2160 4 bounces + mirror", 8192, 3840, 2160, 0, 5, 0b1000, 1, "K1", 48),
    # C5 in the reference's terms: the cross-frame accumulation of svotrace.comp:712-719 over frames 2..65 on a fresh image,
    # one persistent launch (svo_set_sequence); a "frame" of the table is the whole 64-frame sequence
    ("C5 8192^3 1920x1080 64 accumulated frames (svotrace.comp:712-719)", 8192, 1920, 1080, 0, 2, 0, -64, "K1", 32),
    ("C5spp: 64 samples per frame, the library's reading of the dormant SAMPLES loop", 8192, 1920, 1080, 0, 2, 0, 64, "K1", 32),
]
pools = {}
ctx = hiplib.HipContext(0)
ctx.set_pipeline(1)
ctx.set_hit_records(False)
print("| config | ms/frame | Mrays/s | B_alg/ray | alg GB/s | % of 8 TB/s | CPU 1-thread Mrays/s | parity (subsample) |")
print("|---|---|---|---|---|---|---|---|")
for name, n, w, h, mode, bounces, mirror, spp, camname, step in CONFIGS:
    if n not in pools:
        hm, mm = scene.scene_maps(n)
        pools[n] = ctx.pool_download(ctx.build_from_heightmap(hm, mm))   # built on the GPU; host copy for the oracle
        cur = n
    elif cur != n:
        ctx.pool_upload(pools[n])
        cur = n
    pool = pools[n]
    cam = CAMERAS[camname]
    seq = -spp if spp < 0 else 1          # negative spp in the table = frames of a progressive sequence
    spp = 1 if spp < 0 else spp
    ctx.set_progressive(seq > 1)
    ctx.set_sequence(seq, True)
    ctx.resize(w, h)
    ctx.set_camera(cam)
    ctx.set_params(2, mode, 0, 0, bounces, mirror, spp)
    ctx.set_hit_records(True)
    ctx.dispatch()
    got = {"rgba": ctx.read_color(), "depth": ctx.read_depth(), "hits": ctx.read_hits()}
    ctx.set_hit_records(False)
    ms = float(np.median(ctx.time_frames(2, 8 if spp < 8 else 2)))
    # ray / byte counts: the counting pass handles one sample; spp samples differ only in the random seed
    ctx.set_params(2, mode, 0, 0, bounces, mirror, 1)
    ctx.set_progressive(False)
    ctx.set_sequence(1, False)
    st = ctx.count_frame()
    rays = st["rays"] * spp * seq
    alg = (st["alg_bytes"]) * spp * seq + st["pixels"] * 8 * seq
    t0 = time.perf_counter()
    if seq > 1:
        last, nrays = np.zeros((h, w, 4), dtype=np.uint8), 0
        for f in range(2, 2 + seq):
            ref = oracle.render(pool, w, h, cam, f, mode, bounces=bounces, mirror_mask=mirror, xstep=step, ystep=step, last_rgba=last)
            last = ref["rgba"]
            nrays += ref["stats"]["rays"]
        ref["stats"]["rays"] = nrays
    else:
        ref = oracle.render(pool, w, h, cam, 2, mode, bounces=bounces, mirror_mask=mirror, spp=spp, xstep=step, ystep=step)
    dt = time.perf_counter() - t0
    sub = (slice(0, h, step), slice(0, w, step))
    ok = (ref["rgba"][sub] == got["rgba"][sub]).all() and \
        (ref["depth"].view(np.uint32)[sub] == got["depth"].view(np.uint32)[sub]).all() and \
        (ref["hits"]["pointer"][sub] == got["hits"]["pointer"][sub]).all()
    cpu = ref["stats"]["rays"] / dt / 1e6
    print("| %s | %.3f | %.0f | %.0f | %.0f | %.1f | %.2f | %s (%d px) |" % (
        name, ms, rays / ms / 1e3, alg / max(rays, 1), alg / ms / 1e6, alg / ms / 1e6 / 8000 * 100, cpu,
        "bit-exact" if ok else "MISMATCH", ref["stats"]["pixels"]))
