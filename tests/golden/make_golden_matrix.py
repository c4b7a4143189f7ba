#!/usr/bin/env python3
"""The cells of the round-6 measurement matrix (profiles/round6_matrix.md: SURVEY 8(d)'s cameras K0 / K1 / K2 x three scenes)
rendered by the reference shader itself under llvmpipe (oracle/_ref/llvmpipe_ref, shader read from /root/reference at run
time), at the benchmark's full size: 8192^3, 1920x1080, renderMode 0, primary + 1 bounce.  Every STEP-th pixel in x and y of
colour, depth and the first cast's hit record is kept; the pools are not stored (the generators are deterministic; a CRC
catches drift).  Plus the new scene family ("caves", scene/svo_scene.c family 1) at 128^3 and 256^3 and the hostile one ("dust", family 2: floating
particles) at 128^3 with the pools stored, full images, every render mode.  K0 and K1 over the default terrain are already in config3_8192.npz.
Runs only in the build container.   python tests/golden/make_golden_matrix.py"""
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
from svo_raytracer_amd.cameras import CAMERAS, cave_camera  # noqa: E402

CAMERAS = dict(CAMERAS, CAVE=cave_camera(8192, 1, 8, 64))     # inside the largest cave of the caves scene

SHADER = "/root/reference/src/shaders/svotrace.comp"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "llvmpipe_ref")
STEP = 8
# scene key: (family, seed, amp, dens)
SCENES = {"t1a8": ("terrain", 1, 8, 0), "t2a18": ("terrain", 2, 18, 0), "c1a8d64": ("caves", 1, 8, 64), "d1a8x24": ("dust", 1, 8, 24)}
# full size: name, scene, camera, frameNumber, renderMode
FULL = [("t1a8_K2_f2", "t1a8", "K2", 2, 0),
        ("t2a18_K0_f2", "t2a18", "K0", 2, 0), ("t2a18_K1_f2", "t2a18", "K1", 2, 0), ("t2a18_K2_f5", "t2a18", "K2", 5, 0),
        ("c1a8d64_K0_f2", "c1a8d64", "K0", 2, 0), ("c1a8d64_K1_f2", "c1a8d64", "K1", 2, 0), ("c1a8d64_K2_f2", "c1a8d64", "K2", 2, 0),
        ("c1a8d64_K1_m2", "c1a8d64", "K1", 2, 2), ("c1a8d64_CAVE_f2", "c1a8d64", "CAVE", 2, 0),
        ("d1a8x24_K0_f2", "d1a8x24", "K0", 2, 0), ("d1a8x24_K1_f2", "d1a8x24", "K1", 2, 0), ("d1a8x24_K2_f2", "d1a8x24", "K2", 2, 0)]
# small, pools stored: pool key -> (n, seed, amp, dens)
SMALL_POOLS = {"c128": (128, 1, 8, 64), "c256": (256, 2, 8, 128), "d128": (128, 1, 8, 64 << 16)}     # (bits 16..: the dust level's density)


def cam_line(cam):
    return "cam " + " ".join("%08x" % struct.unpack("<I", struct.pack("<f", float(v)))[0] for v in cam)


def llvmpipe(job):
    r = subprocess.run([REF_BIN, SHADER], input=("\n".join(job) + "\n").encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-500:]


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out = {"step": np.array([STEP])}
    tmp = tempfile.mkdtemp(prefix="golden_mx_")
    # ---- the new family, small, every mode, full images, patched == plain checked
    small = []
    for pk, (n, seed, amp, dens) in SMALL_POOLS.items():
        pool, _ = scene.build_scene3(n, seed, amp, dens & 0xffff, dens >> 16)
        out["pool/" + pk] = pool
        path = os.path.join(tmp, pk + ".bin")
        pool.tofile(path)
        job = ["pool " + path]
        cs = [(pk + "_%s_m%d" % (k, m), k, 2 if m != 0 else 3, m) for k in ("K0", "K1", "K2") for m in (0, 1, 2, 3)]
        for name, k, frame, mode in cs:
            job += ["size 128 96", cam_line(CAMERAS[k]), "frame %d" % frame, "mode %d" % mode, "ptrpatch 0",
                    "render " + os.path.join(tmp, name), "ptrpatch 1", "render " + os.path.join(tmp, name + "_p")]
        llvmpipe(job)
        for name, k, frame, mode in cs:
            rgba = np.fromfile(os.path.join(tmp, name + ".rgba"), dtype=np.uint8).reshape(96, 128, 4)
            depth = np.fromfile(os.path.join(tmp, name + ".depth"), dtype=np.uint32).reshape(96, 128)
            rgba_p = np.fromfile(os.path.join(tmp, name + "_p.rgba"), dtype=np.uint8).reshape(96, 128, 4)
            depth_p = np.fromfile(os.path.join(tmp, name + "_p.depth"), dtype=np.uint32).reshape(96, 128)
            ptr = np.fromfile(os.path.join(tmp, name + "_p.ptr"), dtype=np.uint32).reshape(96, 128, 4)
            same = bool((rgba == rgba_p).all() and (depth == depth_p).all())
            out[name + "/rgba"], out[name + "/depth_bits"], out[name + "/first_hit"] = rgba, depth, ptr
            out[name + "/cam"] = np.asarray(CAMERAS[k], dtype=np.float32)
            out[name + "/meta"] = np.array([128, 96, frame, mode, int(same)], dtype=np.int32)
            small.append(name + ":" + pk)
            print("%-16s mode %d hits %5d patched==plain %s" % (name, mode, int((ptr[..., 0] != 0).sum()), same), flush=True)
    out["index_small"] = np.array(small)
    # ---- the matrix cells at full size
    W, H = 1920, 1080
    for sk, (family, seed, amp, dens) in SCENES.items():
        cases = [c for c in FULL if c[1] == sk]
        pool, _ = scene.build(family, 8192, seed, amp, dens, dens)
        path = os.path.join(tmp, sk + ".bin")
        pool.tofile(path)
        out[sk + "/pool_crc32"] = np.array([zlib.crc32(pool.tobytes())], dtype=np.uint32)
        out[sk + "/pool_size"] = np.array([pool.size])
        out[sk + "/scene"] = np.array([family, str(seed), str(amp), str(dens)])
        del pool
        job = ["pool " + path]
        for name, _, k, frame, mode in cases:
            job += ["size %d %d" % (W, H), cam_line(CAMERAS[k]), "frame %d" % frame, "mode %d" % mode, "ptrpatch 1",
                    "render " + os.path.join(tmp, name)]
        llvmpipe(job)
        os.remove(path)
        sub = (slice(0, H, STEP), slice(0, W, STEP))
        for name, _, k, frame, mode in cases:
            p = os.path.join(tmp, name)
            out[name + "/rgba"] = np.fromfile(p + ".rgba", dtype=np.uint8).reshape(H, W, 4)[sub].copy()
            out[name + "/depth_bits"] = np.fromfile(p + ".depth", dtype=np.uint32).reshape(H, W)[sub].copy()
            out[name + "/first_hit"] = np.fromfile(p + ".ptr", dtype=np.uint32).reshape(H, W, 4)[sub].copy()
            out[name + "/cam"] = np.asarray(CAMERAS[k], dtype=np.float32)
            out[name + "/meta"] = np.array([W, H, frame, mode, 2, 0], dtype=np.int32)
            for ext in (".rgba", ".depth", ".ptr"):
                os.remove(p + ext)
            print(name, "hits in the subsample", int((out[name + "/first_hit"][..., 0] != 0).sum()), "of", out[name + "/rgba"].shape[:2], flush=True)
    out["index_full"] = np.array([c[0] + ":" + c[1] for c in FULL])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "matrix_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
