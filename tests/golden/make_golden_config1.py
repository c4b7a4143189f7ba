#!/usr/bin/env python3
"""BASELINE.json config 1: 512^3 procedural SVO, 256x256 frame, primary rays only -- the reference shader
itself under llvmpipe (oracle/_ref/llvmpipe_ref, shader read from /root/reference at run time).  The 5 MB pool
is not stored: the generator is deterministic, a CRC of the pool is kept to catch drift.
Runs only in the build container.   python tests/golden/make_golden_config1.py"""
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


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    pool, _ = scene.build_scene(512)
    tmp = tempfile.mkdtemp(prefix="golden_c1_")
    pool.tofile(os.path.join(tmp, "pool.bin"))
    out = {"pool_crc32": np.array([zlib.crc32(pool.tobytes())], dtype=np.uint32), "pool_size": np.array([pool.size])}
    job = ["pool " + os.path.join(tmp, "pool.bin"), "size 256 256", "frame 2"]
    cases = [("K0", 1), ("K0", 3), ("K1", 1)]   # mode 1 / 3: primary ray only (iteration heat map, normals)
    for camname, mode in cases:
        hexs = " ".join("%08x" % struct.unpack("<I", struct.pack("<f", float(v)))[0] for v in CAMERAS[camname])
        job += ["cam " + hexs, "mode %d" % mode, "ptrpatch 1", "render " + os.path.join(tmp, "%s_m%d" % (camname, mode))]
    r = subprocess.run([REF_BIN, SHADER], input=("\n".join(job) + "\n").encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    for camname, mode in cases:
        p = os.path.join(tmp, "%s_m%d" % (camname, mode))
        out["%s_m%d/rgba" % (camname, mode)] = np.fromfile(p + ".rgba", dtype=np.uint8).reshape(256, 256, 4)
        out["%s_m%d/depth_bits" % (camname, mode)] = np.fromfile(p + ".depth", dtype=np.uint32).reshape(256, 256)
        out["%s_m%d/first_hit" % (camname, mode)] = np.fromfile(p + ".ptr", dtype=np.uint32).reshape(256, 256, 4)
        out["%s_m%d/cam" % (camname, mode)] = np.asarray(CAMERAS[camname], dtype=np.float32)
        print(camname, mode, "hits", int((out["%s_m%d/first_hit" % (camname, mode)][..., 0] != 0).sum()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config1_512.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
