#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE shader
itself (/root/reference/src/shaders/svotrace.comp, read at run time, never copied)
on Mesa llvmpipe through oracle/_ref/llvmpipe_ref.

Runs only in the build container (needs /root/reference and swrast_dri.so).  The
committed outputs are data only: pool bytes, camera floats, and the images the
reference produced for them.

    python tests/golden/make_golden.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import svo_raytracer_amd.scene as scene  # noqa: E402
import poolbuilder  # noqa: E402
from svo_raytracer_amd import hostlib  # noqa: E402
from oracle import octree as restated  # noqa: E402

SHADER = "/root/reference/src/shaders/svotrace.comp"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "llvmpipe_ref")
OUT = os.path.dirname(os.path.abspath(__file__))

L1, L2, R1, R2 = (-1.6, -0.9, -1.0), (-1.6, 0.9, -1.0), (1.6, -0.9, -1.0), (1.6, 0.9, -1.0)


def rot_cam(pos, pitch, yaw):
    """Corner directions rotated by Ry(yaw) * Rx(pitch) in float64, rounded to f32.
    (Not JOML-exact and does not need to be: the 15 floats are stored in the fixture.)"""
    cx, sx, cy, sy = np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    m = ry @ rx
    v = [np.asarray(pos, dtype=np.float64)] + [m @ np.asarray(c) for c in (L1, L2, R1, R2)]
    return np.concatenate(v).astype(np.float32)


def scaled(cam, k):
    """camera position mapped into the [1, 1 + 2^-k]^3 sub-cube of a deep-embedded scene"""
    cam = cam.copy()
    cam[:3] = (1.0 + (cam[:3].astype(np.float64) - 1.0) / (1 << k)).astype(np.float32)
    return cam


K0 = rot_cam((1.5, 1.5, 2.0), 0.0, 0.0)            # reference default, Main.java:120 + Camera.java:13-18
K1 = rot_cam((1.5, 1.42, 1.5), -0.5, 0.3)          # inside the cube, pitched toward the terrain
K2 = rot_cam((1.2, 1.40, 1.8), -0.08, 0.7)         # grazing view, long rays
KOUT = rot_cam((1.45, 1.7, 2.9), -0.35, 0.1)       # outside the cube: world faces -> zero / NaN normals (Q3, Q4)
KDUST = rot_cam((1.3, 1.06, 1.3), -1.2, 0.2)       # looking down at the floor of the dust scene
KEDIT = rot_cam((1.5, 1.62, 1.55), -1.15, 0.4)     # looking down at the SDF-edited area of the 64^3 world


def edited_pool(base):
    """A pool after SDF brush edits through the restatement of Octree.useSDFBrush (oracle/octree_restatement.cpp) (Octree.java:700-885):
    re-tagged interior nodes that keep stale child-pointer / mask bytes (quirk Q5), DELETE_VALUE (127)
    orphans, appended subtrees with SDF normals."""
    o = hostlib.Octree(4096)
    o.adopt(base)
    restated.useSDFBrushSphere(o, (20, 24, 40), 7, 2, worldSize=64, maxLOD=6)      # add material 2
    restated.useSDFBrushSphere(o, (44, 19, 24), 6, 0, worldSize=64, maxLOD=6)      # carve
    restated.useSDFBrushBox(o, (34, 26, 30), 3, 5, 4, 3, worldSize=64, maxLOD=6)   # Main.java:246-248 style box
    restated.useSDFBrushSphere(o, (30, 21, 30), 9, 1, worldSize=64, maxLOD=6)      # big fill: whole sub-trees become tag 2
    return o.getByteBuffer()


def cases():
    s64, _ = scene.build_scene(64)
    s128, _ = scene.build_scene(128)
    s256, _ = scene.build_scene(256)
    red = s64.copy()
    red[0] = 0  # first dword 0 -> red debug square (svotrace.comp:696-700)
    pools = {
        "s64": s64, "s128": s128, "s64red": red,
        "s128k6": scene.embed_deep(s128, 6),   # depth 13: Phong branch res.depth >= 10
        "s256k4": scene.embed_deep(s256, 4),   # depth 12, 0.0625 wide: bounce LOD cap (maxDepth 11), quirk Q2
        # floor + lattice of isolated voxels: shadow rays with > 260 iterations (penumbra branch,
        # svotrace.comp:616-619) and packed-555 normals (quirk Q4 / Q7: NaN rays, iter 1501)
        "dust256": poolbuilder.pool_from_grid(poolbuilder.dust_grid(256, floor=10, cell=8))[0],
        "s64sdf": edited_pool(s64),
    }
    c = []
    for m in (0, 1, 2, 3):
        c.append(("s64_K0_m%d" % m, "s64", 96, 64, K0, 2, m))
        c.append(("s128_K0_m%d" % m, "s128", 128, 96, K0, 2, m))
    for m in (0, 2):
        c.append(("s128_K1_m%d" % m, "s128", 128, 96, K1, 2, m))
        c.append(("s128_K2_m%d" % m, "s128", 128, 96, K2, 2, m))
    for m in (0, 2, 3):
        c.append(("s128_KOUT_m%d" % m, "s128", 96, 56, KOUT, 2, m))
    c.append(("s64_K0_m2_odd", "s64", 100, 60, K0, 2, 2))          # size not a multiple of 8
    c.append(("s64red_K0_m2", "s64red", 64, 48, K0, 2, 2))
    for f in (3, 7, 99):
        c.append(("s128_K1_m0_f%d" % f, "s128", 96, 64, K1, f, 0))  # rand() seeds
    for m in (0, 2):
        c.append(("s128k6_K0_m%d" % m, "s128k6", 128, 96, scaled(K0, 6), 2, m))
        c.append(("s128k6_K1_m%d" % m, "s128k6", 128, 96, scaled(K1, 6), 2, m))
        c.append(("s256k4_K1_m%d" % m, "s256k4", 128, 96, scaled(K1, 4), 2, m))
    c.append(("s256k4_K2_m0", "s256k4", 128, 96, scaled(K2, 4), 2, 0))
    for m in (0, 1, 2, 3):
        c.append(("dust256_KDUST_m%d" % m, "dust256", 64, 48, KDUST, 2, m))
    c.append(("dust256_K1_m2", "dust256", 96, 64, K1, 2, 2))
    for m in (0, 1, 2, 3):
        c.append(("s64sdf_KEDIT_m%d" % m, "s64sdf", 96, 64, KEDIT, 2, m))
    c.append(("s64sdf_K0_m2", "s64sdf", 96, 64, K0, 2, 2))
    return pools, c


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    pools, cs = cases()
    tmp = tempfile.mkdtemp(prefix="golden_")
    job = []
    cur_pool = None
    for name, pk, w, h, cam, frame, mode in cs:
        if pk != cur_pool:
            path = os.path.join(tmp, pk + ".bin")
            pools[pk].tofile(path)
            job.append("pool " + path)
            cur_pool = pk
        hexs = " ".join("%08x" % struct.unpack("<I", struct.pack("<f", float(v)))[0] for v in cam)
        job += ["size %d %d" % (w, h), "cam " + hexs, "frame %d" % frame, "mode %d" % mode,
                "ptrpatch 0", "render " + os.path.join(tmp, name),
                "ptrpatch 1", "render " + os.path.join(tmp, name + "_p")]
    r = subprocess.run([REF_BIN, SHADER], input=("\n".join(job) + "\n").encode(), capture_output=True)
    sys.stderr.write(r.stderr.decode()[-600:])
    assert r.returncode == 0
    out = {}
    index = []
    for name, pk, w, h, cam, frame, mode in cs:
        rgba = np.fromfile(os.path.join(tmp, name + ".rgba"), dtype=np.uint8).reshape(h, w, 4)
        depth = np.fromfile(os.path.join(tmp, name + ".depth"), dtype=np.uint32).reshape(h, w)
        rgba_p = np.fromfile(os.path.join(tmp, name + "_p.rgba"), dtype=np.uint8).reshape(h, w, 4)
        depth_p = np.fromfile(os.path.join(tmp, name + "_p.depth"), dtype=np.uint32).reshape(h, w)
        ptr = np.fromfile(os.path.join(tmp, name + "_p.ptr"), dtype=np.uint32).reshape(h, w, 4)
        # the instrumented program must not change what the unmodified one computes
        same = (rgba == rgba_p).all() and (depth == depth_p).all()
        print("%-22s %4dx%-4d mode %d frame %2d hits %6d  patched==plain %s" %
              (name, w, h, mode, frame, int((ptr[..., 0] != 0).sum()), same))
        out[name + "/rgba"] = rgba
        out[name + "/depth_bits"] = depth
        out[name + "/first_hit"] = ptr  # pointer, value, leafMask field, (level << 16) | iter of the first cast
        out[name + "/cam"] = np.asarray(cam, dtype=np.float32)
        out[name + "/meta"] = np.array([w, h, frame, mode, int(same)], dtype=np.int32)
        index.append(name + ":" + pk)
    for pk, p in pools.items():
        out["pool/" + pk] = p
    out["index"] = np.array(index)
    np.savez_compressed(os.path.join(OUT, "llvmpipe_golden.npz"), **out)
    print("wrote", os.path.join(OUT, "llvmpipe_golden.npz"),
          os.path.getsize(os.path.join(OUT, "llvmpipe_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
