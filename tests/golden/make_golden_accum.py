#!/usr/bin/env python3
"""Golden vectors for the reference's dormant cross-frame accumulation (svotrace.comp:712-719:
finalcolor = (frameNumber * lastcolor + finalcolor) / (frameNumber + 1), frozen from MAX_FRAME_ITER on).
The block is commented out in the live shader; oracle/_ref/llvmpipe_ref switches it on IN MEMORY (by line number, the
file is never copied or edited) and keeps the images from render to render as Main.java does.  Sequences of consecutive
frameNumbers on one persistent framebuffer, the reference shader itself on llvmpipe.

    python tests/golden/make_golden_accum.py      (build container only)
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import svo_raytracer_amd.scene as scene  # noqa: E402
from svo_raytracer_amd.cameras import CAMERAS  # noqa: E402

SHADER = "/root/reference/src/shaders/svotrace.comp"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "llvmpipe_ref")
OUT = os.path.dirname(os.path.abspath(__file__))

# name, pool size, W, H, camera, mode, frame numbers rendered in this order onto one framebuffer
SEQUENCES = [
    ("gi_K1", 128, 96, 64, "K1", 0, [2, 3, 4, 5, 6, 7]),          # the use case: GI converging over frames
    ("shadow_K0", 64, 64, 48, "K0", 2, [2, 3, 4]),
    ("gi_freeze", 64, 64, 48, "K1", 0, [98, 99, 100, 101]),       # MAX_FRAME_ITER 100: from frame 100 on the image is frozen
]


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    tmp = tempfile.mkdtemp(prefix="accum_")
    job, pools = [], {}
    for name, n, w, h, cam, mode, frames in SEQUENCES:
        if n not in pools:
            pools[n] = scene.build_scene(n)[0]
            pools[n].tofile(os.path.join(tmp, "p%d.bin" % n))
        hexs = " ".join("%08x" % struct.unpack("<I", struct.pack("<f", float(v)))[0] for v in CAMERAS[cam])
        job += ["pool " + os.path.join(tmp, "p%d.bin" % n), "size %d %d" % (w, h), "cam " + hexs, "mode %d" % mode,
                "accum 1", "fresh", "keep 1"]
        for i, f in enumerate(frames):
            job += ["frame %d" % f, "render " + os.path.join(tmp, "%s_%d" % (name, i))]
        job += ["fresh", "keep 0"]
    r = subprocess.run([REF_BIN, SHADER], input=("\n".join(job) + "\n").encode(), capture_output=True)
    sys.stderr.write(r.stderr.decode()[-400:])
    assert r.returncode == 0
    out, index = {}, []
    for name, n, w, h, cam, mode, frames in SEQUENCES:
        rgba = np.stack([np.fromfile(os.path.join(tmp, "%s_%d.rgba" % (name, i)), dtype=np.uint8).reshape(h, w, 4)
                         for i in range(len(frames))])
        depth = np.stack([np.fromfile(os.path.join(tmp, "%s_%d.depth" % (name, i)), dtype=np.uint32).reshape(h, w)
                          for i in range(len(frames))])
        out[name + "/rgba"] = rgba
        out[name + "/depth_bits"] = depth
        out[name + "/cam"] = np.asarray(CAMERAS[cam], dtype=np.float32)
        out[name + "/meta"] = np.array([n, w, h, mode] + frames, dtype=np.int32)
        index.append(name)
        print(name, rgba.shape, "distinct frames:", len({rgba[i].tobytes() for i in range(len(frames))}))
    out["index"] = np.array(index)
    np.savez_compressed(os.path.join(OUT, "accum_golden.npz"), **out)
    print("wrote accum_golden.npz", os.path.getsize(os.path.join(OUT, "accum_golden.npz")))


if __name__ == "__main__":
    main()
