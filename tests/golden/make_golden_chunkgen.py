#!/usr/bin/env python3
"""The world builder's first stage under the reference's own shader: src/shaders/chunkgen-heightmap.comp turns a 16-bit
height map and a material map into a dense chunk of voxels (solid up to int(r / 65536.0 * 2048), the top five layers the
material map's value, value 1 below; chunkgen-heightmap.comp:13-31), dispatched by Octree.constructCompleteOctree chunk by
chunk (Octree.java:274-287).  oracle/_ref/llvmpipe_ref runs that shader itself (read from /root/reference at run time, never
copied) with the reference's bindings and upload formats, on seeded maps:

  A  a 128 x 128 terrain whose heights stay below 128 voxels: the whole 128^3 world as one chunk, and a 64^3 chunk at an offset
  B  128 x 128 raw values over the full 16-bit range (0 .. 65535 -> 0 .. 2047 voxels), materials over all 256 byte values:
     64^3 chunks at y offsets that cut the columns (what the height scaling, the signed r8i material image and the <= 4 test
     really do)
  C  128 x 128 random columns below 128 voxels with every non-zero material byte: the whole 128^3 world (the GPU builder's
     two entry points must agree on it: svo_build_from_voxels on these voxels, svo_build_from_heightmap16 on the maps)

    python tests/golden/make_golden_chunkgen.py        (build container only)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHADER = "/root/reference/src/shaders/chunkgen-heightmap.comp"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "llvmpipe_ref")
N = 128
# name, chunk size, origin x y z
CHUNKS_A = [("A_world", 128, 0, 0, 0), ("A_part", 64, 64, 32, 64)]
CHUNKS_C = [("C_world", 128, 0, 0, 0)]
CHUNKS_B = [("B_low", 64, 0, 0, 0), ("B_mid", 64, 64, 992, 64), ("B_high", 64, 32, 1984, 16), ("B_edge", 64, 64, 2016, 0)]


def maps_a(rng):
    x = np.arange(N)
    base = 40 + 25 * np.sin(x[None, :] / 9.0) + 20 * np.cos(x[:, None] / 13.0)
    h = np.clip(base + rng.integers(-3, 4, size=(N, N)), 0, 127).astype(np.int64)
    raw = (h * 32 + rng.integers(0, 32, size=(N, N))).astype(np.uint16)      # any raw value in [32 h, 32 h + 31] is h voxels
    mat = rng.integers(1, 4, size=(N, N)).astype(np.uint8)
    return raw, mat


def maps_b(rng):
    raw = rng.integers(0, 65536, size=(N, N)).astype(np.uint16)
    raw[0, :8] = [0, 31, 32, 65535, 65504, 65503, 1, 33]                       # the ends of the scale
    raw[64:96, 64:96] = (32 * (990 + rng.integers(0, 70, size=(32, 32)))).astype(np.uint16)   # columns that end inside B_mid
    raw[16:48, 32:64] = (32 * (1980 + rng.integers(0, 68, size=(32, 32)))).astype(np.uint16)  # ... and inside B_high
    mat = rng.integers(0, 256, size=(N, N)).astype(np.uint8)                   # incl. 0 and bytes >= 128 (negative in r8i)
    return raw, mat


def maps_c(rng):
    raw = rng.integers(0, 4096, size=(N, N)).astype(np.uint16)
    mat = rng.integers(1, 256, size=(N, N)).astype(np.uint8)
    return raw, mat


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    rng = np.random.default_rng(20260401)
    tmp = tempfile.mkdtemp(prefix="golden_chunkgen_")
    out, job, index = {}, [], []
    for tag, (raw, mat), chunks in (("A", maps_a(rng), CHUNKS_A), ("B", maps_b(rng), CHUNKS_B), ("C", maps_c(rng), CHUNKS_C)):
        raw.tofile(os.path.join(tmp, tag + ".h16"))
        mat.tofile(os.path.join(tmp, tag + ".mat"))
        out[tag + "/raw"], out[tag + "/mat"] = raw, mat
        job.append("maps %d %s %s" % (N, os.path.join(tmp, tag + ".h16"), os.path.join(tmp, tag + ".mat")))
        for name, c, ox, oy, oz in chunks:
            job.append("chunk %d %d %d %d %s" % (c, ox, oy, oz, os.path.join(tmp, name + ".vox")))
            index.append(name)
            out[name + "/meta"] = np.array([c, ox, oy, oz], dtype=np.int32)
    r = subprocess.run([REF_BIN, SHADER, "chunkgen"], input=("\n".join(job) + "\n").encode(), capture_output=True)
    sys.stderr.write(r.stderr.decode()[-600:])
    assert r.returncode == 0
    for name in index:
        c = int(out[name + "/meta"][0])
        out[name + "/voxels"] = np.fromfile(os.path.join(tmp, name + ".vox"), dtype=np.uint8).reshape(c, c, c)   # [z][y][x]
        v = out[name + "/voxels"]
        print(name, "solid", int((v != 0).sum()), "of", v.size, "values", np.unique(v)[:12])
    out["index"] = np.array(index)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chunkgen_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
