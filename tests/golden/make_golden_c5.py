#!/usr/bin/env python3
"""BASELINE.json config 5 in the reference's own terms: "64 spp accumulated path-traced GI" at 8192^3, 1920x1080.

The only accumulation the reference contains is the cross-frame blend of svotrace.comp:712-719,
    finalcolor = (frameNumber * lastcolor + finalcolor) / (frameNumber + 1)        (frozen from MAX_FRAME_ITER = 100 on)
read back through the rgba8 image from frame to frame (its SAMPLES loop, :668-670, is commented out around a body that
seeds every sample alike).  Main.java pre-increments frameNumber from 1 (:16, :275): a camera at rest renders frameNumber
2, 3, 4, ...  So 64 accumulated samples = the 64 frames 2..65 onto ONE persistent framebuffer.

This script lets the reference shader itself do that under llvmpipe (oracle/_ref/llvmpipe_ref switches the commented
block on IN MEMORY, by line number; the file is never copied or edited; images are kept from render to render as
Main.java keeps them): 8192^3 bench scene, camera K1, renderMode 0, 64 x 1.4 s.  Every STEP-th pixel in x and y of colour
and depth of frames 2, 3, 33 and 65 is kept (the pool is regenerated, its CRC is in config3_8192.npz).

    python tests/golden/make_golden_c5.py        (build container only; ~2.5 min)
"""
import os
import struct
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import svo_raytracer_amd.scene as scene  # noqa: E402
from svo_raytracer_amd.cameras import CAMERAS  # noqa: E402

SHADER = "/root/reference/src/shaders/svotrace.comp"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "llvmpipe_ref")
STEP = 8
N, W, H, CAM, MODE = 8192, 1920, 1080, "K1", 0
FRAMES = list(range(2, 66))       # 64 frames
KEEP = [2, 3, 33, 65]


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    pool, _ = scene.build_scene(N)
    tmp = tempfile.mkdtemp(prefix="golden_c5_")
    pool.tofile(os.path.join(tmp, "pool.bin"))
    hexs = " ".join("%08x" % struct.unpack("<I", struct.pack("<f", float(v)))[0] for v in CAMERAS[CAM])
    job = ["pool " + os.path.join(tmp, "pool.bin"), "size %d %d" % (W, H), "cam " + hexs, "mode %d" % MODE, "accum 1", "fresh",
           "keep 1"]
    for f in FRAMES:
        job += ["frame %d" % f, "render " + os.path.join(tmp, "f%d" % f if f in KEEP else "scratch")]
    job += ["fresh", "keep 0"]
    r = subprocess.run([REF_BIN, SHADER], input=("\n".join(job) + "\n").encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    out = {"step": np.array([STEP]), "frames": np.array(FRAMES, dtype=np.int32), "keep": np.array(KEEP, dtype=np.int32),
           "meta": np.array([N, W, H, MODE], dtype=np.int32), "cam": np.asarray(CAMERAS[CAM], dtype=np.float32),
           "pool_crc32": np.array([zlib.crc32(pool.tobytes())], dtype=np.uint32), "pool_size": np.array([pool.size])}
    sub = (slice(0, H, STEP), slice(0, W, STEP))
    for f in KEEP:
        p = os.path.join(tmp, "f%d" % f)
        out["f%d/rgba" % f] = np.fromfile(p + ".rgba", dtype=np.uint8).reshape(H, W, 4)[sub].copy()
        out["f%d/depth_bits" % f] = np.fromfile(p + ".depth", dtype=np.uint32).reshape(H, W)[sub].copy()
        print("frame", f, "mean rgb", out["f%d/rgba" % f][..., :3].mean(axis=(0, 1)).round(2))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_progressive.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
