#!/usr/bin/env python3
"""Golden vectors for pools no builder produces, again from the REFERENCE shader itself under Mesa llvmpipe
(oracle/_ref/llvmpipe_ref; the shader is read from /root/reference at run time, never copied):

  * random structurally valid pools (tests/fuzzpool.py): arbitrary tag mixes, empty interior nodes, interior nodes
    without a child block, any 16-bit 'normal', tag-2 nodes with stale bytes, DELETE_VALUE, materials the shader has no
    colour for (renderMode 2 then reads a variable it never set); renderModes 4, 5 and -1 on two of them;
  * the same pools with child pointers re-aimed: backwards (cycles: the cast runs into the iteration cap), into the
    middle of other records (any alignment), and past the last byte in use -- inside the zero bytes the reference's
    buffer always has behind memOffset (its byte[] is far larger than the tree; the harness is told to keep 1 MiB,
    and only such pointers are generated: what GL does beyond a buffer's end is not the reference's behaviour).

  * the terrain scene at frame numbers up to +-2^31, and llvmpipe's sin / cos themselves at such arguments (a probe
    shader of our own, tools/probes/sin_probe.comp: mode 0 sin, 1 cos of frame + 977 i, 2 sin(12.9898 i + 7.8233 frame)).

Runs only in the build container.  Output: tests/golden/fuzz_golden.npz (data only).

  * hostile cameras and the smallest legal pools (hs_*); degenerate image sizes (sz_*).

    python tests/golden/make_golden_fuzz.py
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
from svo_raytracer_amd.cameras import CAMERAS, rot_cam  # noqa: E402
import fuzzpool  # noqa: E402

SHADER = "/root/reference/src/shaders/svotrace.comp"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "llvmpipe_ref")
OUT = os.path.dirname(os.path.abspath(__file__))
PAD = 1 << 20


def size_of(name):
    if name.startswith("sz_"):                      # degenerate image sizes: sz_<W>x<H>_...
        w, h = name.split("_")[1].split("x")
        return int(w), int(h)
    return (40, 24) if name.startswith("hs_") else (56, 36)



def interior_offsets(p, limit=4000):
    out, stack = [], [0]
    while stack and len(out) < limit:
        o = stack.pop()
        cp = int.from_bytes(bytes(p[o + 1:o + 5]), "big", signed=True)
        if cp == 0:
            continue
        out.append(o)
        mask = (int(p[o + 5]) << 8) | int(p[o + 6])
        c = o + cp
        for n in range(8):
            tag = (mask >> (2 * n)) & 3
            if tag == 0 and c + 7 <= p.size:
                stack.append(c)
            c += {0: 7, 1: 3, 2: 7, 3: 1}[tag]
    return out


def mangled(pool, seed):
    """child pointers re-aimed; every target stays below len + PAD - 64 (inside the buffer the reference would have)"""
    rng = np.random.RandomState(7000 + seed)
    p = pool.copy()
    inter = interior_offsets(p)
    for o in rng.choice(inter[1:], size=min(10, len(inter) - 1), replace=False):
        kind = rng.randint(0, 3)
        if kind == 0:
            target = 0                                             # back to the root block: a cycle
        elif kind == 1:
            target = int(p.size) + int(rng.randint(0, PAD - 4096))  # zero bytes behind the tree
        else:
            target = int(rng.randint(7, p.size))                   # somewhere inside the pool, any alignment
        p[o + 1:o + 5] = np.frombuffer(int(target - int(o)).to_bytes(4, "big", signed=True), dtype=np.uint8)
    return p


def cases():
    cams = {"K0": CAMERAS["K0"], "K1": CAMERAS["K1"], "KA": rot_cam((1.45, 1.7, 2.9), -0.35, 0.1), "KB": rot_cam((1.5, 1.5, 1.5), 0.9, 2.0)}
    pools, cs = {}, []
    for seed in range(6):
        base = fuzzpool.random_pool(seed, max_depth=5 + seed % 2, p_interior=0.8, p_empty=0.8)
        deep = seed % 3 == 2
        if deep:
            base = scene.embed_deep(base, 7)       # depth 12-13: LOD cap on bounce rays, Phong branch
        pools["f%d" % seed] = base
        pools["m%d" % seed] = mangled(base, seed)
        for kind in ("f", "m"):
            for cn, cam in cams.items():
                cam = cam.copy()
                if deep:
                    cam[:3] = (1.0 + (cam[:3].astype(np.float64) - 1.0) / 128).astype(np.float32)
                for mode in (0, 1, 2, 3):
                    if kind == "m" and cn in ("K1", "KA") and mode in (1, 3):
                        continue                   # keep the fixture small
                    cs.append(("%s%d_%s_m%d" % (kind, seed, cn, mode), "%s%d" % (kind, seed), cam, 2 + seed, mode))
            if kind == "f" and seed in (0, 3):     # renderMode 4 returns an unset variable, 5.. and negatives fall off the end of trace()
                for mode in (4, 5, -1):
                    cs.append(("f%d_K0_m%d" % (seed, mode), "f%d" % seed, cams["K0"].copy(), 2 + seed, mode))
    # frameNumber far beyond the goldens of round 1 (2, 3, 7, 99): svotrace.comp:486 feeds frameNumber * 7.8 to sin();
    # beyond 2^23 llvmpipe's range reduction breaks down and the clamp of its result shows, beyond 2.7e8 the float -> int
    # conversion inside it overflows (x86 semantics)
    pools["s128"] = scene.build_scene(128)[0]
    for fr in (-5, 1000000, 16777216, 123456789, 300000000, 2147483647, -2147483648):
        cs.append(("s128_K1_m0_f%d" % fr, "s128", cams["K1"].copy(), fr, 0))
    # hostile cameras on the terrain scene (NaN / infinite / huge / denormal components, zero-length and identical corner
    # rays, the camera on faces, corners and cell boundaries of the cube, far outside, inside solid voxels) and the
    # smallest legal pools; smaller images (HW, HH) -- most of these frames are flat
    for k, cam in enumerate(hostile_cameras(np.array(CAMERAS["K1"], dtype=np.float32))):
        for mode in (0, 1, 2, 3):
            cs.append(("hs_s128_c%d_m%d" % (k, mode), "s128", cam, 2 + k, mode))
    surf = [1, 0, 0, 0, 7, 0x55, 0x55]
    for i in range(8):
        surf += [1 + i % 3, (455 + 20 * i) & 0xff, (455 + 20 * i) >> 8]
    tiny = {"t_root": [1, 0, 0, 0, 0, 0, 0], "t_loop": [1, 0, 0, 0, 0, 0, 0, 0],
            "t_leaves": [1, 0, 0, 0, 7, 0xff, 0xff] + [1, 0, 2, 0, 3, 0, 0, 5], "t_surf": surf}
    for pk, b in tiny.items():
        pools[pk] = np.array(b, dtype=np.uint8)
        for cn in ("K0", "K1", "K2"):
            for mode in (0, 1, 2, 3):
                cs.append(("hs_%s_%s_m%d" % (pk, cn, mode), pk, np.array(CAMERAS[cn], dtype=np.float32), 2, mode))
    # path options the shader carries but does not run, switched on in memory by the harness (`bounces n`: the literal
    # bound of the path loop, svotrace.comp:444; `mirror 1`: the commented-out material test of :500-504 -- value 1
    # scatters, every other value reflects = mirror_mask 0xfffffffd here)
    pools["f1d6"] = fuzzpool.random_pool(1, max_depth=6, p_interior=0.8, p_empty=0.8)
    for pk in ("s128", "f1d6"):
        for bounces, mirror in ((1, 0), (3, 0), (5, 0), (2, 1), (4, 1)):
            for cn in ("K1", "K0"):
                for fr in (2, 9):
                    cs.append(("pv_%s_b%d_r%d_%s_f%d" % (pk, bounces, mirror, cn, fr), pk, cams[cn].copy(), fr, 0,
                               dict(bounces=bounces, mirror=mirror)))
    # degenerate image sizes (1x1, single rows and columns, sizes around the 8-pixel work group)
    pools["s64"] = scene.build_scene(64)[0]
    for w, h in ((1, 1), (1, 37), (53, 1), (3, 3), (4, 4), (5, 9), (7, 8), (8, 7), (9, 9), (15, 17), (64, 3), (2, 130), (1000, 2), (3, 777)):
        cs.append(("sz_%dx%d_K1_m0" % (w, h), "s64", cams["K1"].copy(), 2, 0))
        cs.append(("sz_%dx%d_K0_m2" % (w, h), "s64", cams["K0"].copy(), 3, 2))
    return pools, cs


def hostile_cameras(base):
    cams = []
    for idx, val in ((0, np.nan), (1, np.inf), (2, -np.inf), (4, np.nan), (7, np.inf), (9, 0.0), (13, -0.0), (5, 1e30), (3, 1e-30),
                     (0, 1e30), (1, -1e30), (6, 1e-45), (10, 3e38)):
        c = base.copy()
        c[idx] = val
        cams.append(c)
    z = base.copy(); z[3:] = 0.0; cams.append(z)                                       # normalize(0) = NaN
    n = base.copy(); n[:] = np.nan; cams.append(n)
    for d in ((0.0, -1.0, 0.0), (1.0, 0.0, 0.0), (-1.0, -1.0, -1.0)):                  # one direction for every pixel
        a = base.copy(); a[3:6] = a[6:9] = a[9:12] = a[12:15] = d; cams.append(a)
    for pos in ((1.0, 1.5, 1.5), (2.0, 2.0, 2.0), (1.5, 1.5, 1.5), (1.25, 1.5, 1.75), (40.0, 30.0, -25.0), (-1e6, 1.5, 1.5),
                (1.5, 1.01, 1.5)):
        f = base.copy(); f[:3] = pos; cams.append(f)
    return cams


BIG_FRAMES = (0, -7, 1000000, 8000000, 16777216, 60000000, 100000000, 215000000, 216000000, 1000000000, 2147483647, -2147483648)
PW, PH = 32, 16


def sin_probe(tmp, out):
    """llvmpipe's sin / cos at large arguments, through a probe shader of our own (tools/probes/sin_probe.comp)"""
    job = ["size %d %d" % (PW, PH)]
    for fr in BIG_FRAMES:
        for m in (0, 1, 2):
            job += ["frame %d" % fr, "mode %d" % m, "ptrpatch 0", "render " + os.path.join(tmp, "sp%d_%d" % (fr, m))]
    r = subprocess.run([REF_BIN, os.path.join(ROOT, "tools", "probes", "sin_probe.comp"), "raw"],
                       input=("\n".join(job) + "\n").encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    idx = np.arange(PW * PH, dtype=np.float32)
    for fr in BIG_FRAMES:
        f = np.float32(fr)
        for m in (0, 1, 2):
            if m < 2:
                x = (f + (idx * np.float32(977.0)).astype(np.float32)).astype(np.float32)
            else:
                x = ((idx * np.float32(12.9898)).astype(np.float32) + np.float32(f * np.float32(7.8233))).astype(np.float32)
            out["sinprobe/%d_%d/x" % (fr, m)] = x
            out["sinprobe/%d_%d/ref_bits" % (fr, m)] = np.fromfile(os.path.join(tmp, "sp%d_%d.depth" % (fr, m)), dtype=np.uint32)
    out["sinprobe/index"] = np.array(["%d_%d" % (fr, m) for fr in BIG_FRAMES for m in (0, 1, 2)])


def _bits(v):
    return struct.unpack("<I", struct.pack("<f", v))[0]


FN_BASES = {   # renderMode of tools/probes/fn_probe.comp -> bit patterns the 2 048 consecutive arguments start from
    0: [_bits(1.0) - 1000, _bits(-1.0) - 1000, _bits(0.0), 0x7f800000 - 1000, 0xff800000 - 1000, _bits(0.99), _bits(0.5) - 1000,
        _bits(-0.3), 0x80000000, _bits(2.0), _bits(0.92) - 1000],
    1: [_bits(0.0), _bits(1.0), _bits(87.0), _bits(126.0) - 500, _bits(87.33) - 1000, _bits(88.0), _bits(130.0), 0x7f800000 - 1000,
        _bits(-1.0), _bits(-87.0), _bits(-88.5) - 1000, _bits(-88.72) - 1000, _bits(-126.0) - 500, _bits(-200.0), 1, 0x80000000,
        0xff800000 - 1000, _bits(-89.0), _bits(0.3), _bits(3.0)],
}
FW, FH = 64, 32


def fn_probe(tmp, out):
    """llvmpipe's acos() and the fog term exp(-0.5 * x * 2) at chosen bit patterns (tools/probes/fn_probe.comp)"""
    job = ["size %d %d" % (FW, FH)]
    for m, bl in FN_BASES.items():
        for b in bl:
            job += ["frame %d" % (b if b < 0x80000000 else b - (1 << 32)), "mode %d" % m, "ptrpatch 0",
                    "render " + os.path.join(tmp, "fp%d_%08x" % (m, b))]
    r = subprocess.run([REF_BIN, os.path.join(ROOT, "tools", "probes", "fn_probe.comp"), "raw"],
                       input=("\n".join(job) + "\n").encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    keys = []
    for m, bl in FN_BASES.items():
        for b in bl:
            k = "%d_%08x" % (m, b)
            out["fnprobe/%s/x_bits" % k] = (np.uint32(b) + np.arange(FW * FH, dtype=np.uint32)).astype(np.uint32)
            out["fnprobe/%s/ref_bits" % k] = np.fromfile(os.path.join(tmp, "fp" + k + ".depth"), dtype=np.uint32)
            keys.append(k)
    out["fnprobe/index"] = np.array(keys)


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    pools, cs = cases()
    tmp = tempfile.mkdtemp(prefix="golden_fuzz_")
    job, cur = ["pad %d" % PAD], None
    variant = (2, 0)
    for name, pk, cam, frame, mode, *opt in cs:
        want = (opt[0]["bounces"], opt[0]["mirror"]) if opt else (2, 0)
        if want != variant:
            job += ["bounces %d" % want[0], "mirror %d" % want[1]]
            variant = want
        job.append("size %d %d" % size_of(name))
        if pk != cur:
            path = os.path.join(tmp, pk + ".bin")
            pools[pk].tofile(path)
            job.append("pool " + path)
            cur = pk
        hexs = " ".join("%08x" % struct.unpack("<I", struct.pack("<f", float(v)))[0] for v in cam)
        job += ["cam " + hexs, "frame %d" % frame, "mode %d" % mode, "ptrpatch 0", "render " + os.path.join(tmp, name),
                "ptrpatch 1", "render " + os.path.join(tmp, name + "_p")]
    r = subprocess.run([REF_BIN, SHADER], input=("\n".join(job) + "\n").encode(), capture_output=True)
    sys.stderr.write(r.stderr.decode()[-600:])
    assert r.returncode == 0
    out, index = {}, []
    for name, pk, cam, frame, mode, *opt in cs:
        W, H = size_of(name)
        rgba = np.fromfile(os.path.join(tmp, name + ".rgba"), dtype=np.uint8).reshape(H, W, 4)
        depth = np.fromfile(os.path.join(tmp, name + ".depth"), dtype=np.uint32).reshape(H, W)
        rgba_p = np.fromfile(os.path.join(tmp, name + "_p.rgba"), dtype=np.uint8).reshape(H, W, 4)
        depth_p = np.fromfile(os.path.join(tmp, name + "_p.depth"), dtype=np.uint32).reshape(H, W)
        ptr = np.fromfile(os.path.join(tmp, name + "_p.ptr"), dtype=np.uint32).reshape(H, W, 4)
        same = bool((rgba == rgba_p).all() and (depth == depth_p).all())
        print("%-14s mode %d frame %d hits %5d capped %4d patched==plain %s" %
              (name, mode, frame, int((ptr[..., 0] != 0).sum()), int(((ptr[..., 3] & 0xffff) > 1500).sum()), same))
        out[name + "/rgba"] = rgba
        out[name + "/depth_bits"] = depth
        out[name + "/first_hit"] = ptr
        out[name + "/cam"] = np.asarray(cam, dtype=np.float32)
        out[name + "/meta"] = np.array([W, H, frame, mode, int(same)], dtype=np.int32)
        if opt:   # path options: segments, mirror mask
            out[name + "/path"] = np.array([opt[0]["bounces"], 0xfffffffd if opt[0]["mirror"] else 0], dtype=np.uint32)
        index.append(name + ":" + pk)
    for pk, p in pools.items():
        out["pool/" + pk] = p
    out["index"] = np.array(index)
    sin_probe(tmp, out)
    fn_probe(tmp, out)
    path = os.path.join(OUT, "fuzz_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(cs), "cases")


if __name__ == "__main__":
    main()
