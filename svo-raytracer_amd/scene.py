"""ctypes binding of the procedural scene generator (scene/svo_scene.c)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "scene", "libsvoscene.so")
_lib = None


class SceneStats(ctypes.Structure):
    _fields_ = [
        ("bytes", ctypes.c_uint64),
        ("interior", ctypes.c_uint64),
        ("surface_leaf", ctypes.c_uint64),
        ("nonsurface_leaf", ctypes.c_uint64),
        ("subdiv_leaf", ctypes.c_uint64),
        ("depth", ctypes.c_int32),
        ("hmin", ctypes.c_int32),
        ("hmax", ctypes.c_int32),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = ctypes.CDLL(_LIB_PATH)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.svo_scene_build.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(u8p),
                                      ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(SceneStats)]
        L.svo_scene_build.restype = ctypes.c_int
        L.svo_scene_free.argtypes = [u8p]
        L.svo_scene_free.restype = None
        L.svo_scene_height.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.svo_scene_height.restype = ctypes.c_int
        L.svo_scene_maps.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.svo_scene_maps.restype = ctypes.c_int
        L.svo_pool_validate.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(SceneStats),
                                        ctypes.POINTER(ctypes.c_int)]
        L.svo_pool_validate.restype = ctypes.c_int
        L.svo_scene_build3.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(u8p),
                                       ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(SceneStats)]
        L.svo_scene_build3.restype = ctypes.c_int
        L.svo_scene3_voxels.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.svo_scene3_voxels.restype = ctypes.c_int
        L.svo_scene3_ball_counts.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.svo_scene3_ball_counts.restype = ctypes.c_int
        _lib = L
    return _lib


def build_scene(n, seed=1, amp=8):
    """Build an n^3 procedural terrain SVO. Returns (pool: np.uint8[len], stats dict)."""
    L = lib()
    p = ctypes.POINTER(ctypes.c_uint8)()
    ln = ctypes.c_uint64()
    st = SceneStats()
    rc = L.svo_scene_build(int(n), int(seed), int(amp), ctypes.byref(p), ctypes.byref(ln), ctypes.byref(st))
    if rc != 0:
        raise RuntimeError(f"svo_scene_build({n}) failed rc={rc} (3 = pool would exceed 2^31 bytes: {ln.value})")
    try:
        pool = np.ctypeslib.as_array(p, shape=(ln.value,)).copy()
    finally:
        L.svo_scene_free(p)
    return pool, st.as_dict()


# Scene families (bench.py --scene, SURVEY 8d "Synthetic scenes"): "terrain" = the height field alone (what rounds 1-5
# measured), "caves" = the same terrain under levels of hashed balls that carve it or float over it (scene/svo_scene.c,
# "family 1": craters and cave mouths under overhanging rims, boulders, arches, floating debris -- real 3-D structure,
# the integer analogue of the Worley part of the reference's chunkgen.comp:228-233).
CAVES_DENS = 64     # dens / 256 = probability that a cell next to a surface holds a ball: 2.04 GB at 8192^3 (limit 2^31)
# "dust" (family 2): the terrain under a field of floating particles (balls of radius 1 or 2 in cells of edge N / 256) -- the
# hostile case of an octree walk: rays do not hit the particles, they descend into every coarse cell that holds one.
DUST_DENS = 24      # dust / 256 = probability that an air cell of the dust level holds a particle


def _d(dens, dust):
    return int(dens) | (int(dust) << 16)      # (the C entry points take the dust level's density in bits 16.. of `dens`)


def build_scene3(n, seed=1, amp=8, dens=CAVES_DENS, dust=0):
    """Build an n^3 family-1 ("caves": dens > 0) / family-2 ("dust": dust > 0) SVO.  Returns (pool: np.uint8[len], stats dict)."""
    L = lib()
    p = ctypes.POINTER(ctypes.c_uint8)()
    ln = ctypes.c_uint64()
    st = SceneStats()
    rc = L.svo_scene_build3(int(n), int(seed), int(amp), _d(dens, dust), ctypes.byref(p), ctypes.byref(ln), ctypes.byref(st))
    if rc != 0:
        raise RuntimeError(f"svo_scene_build3({n}) failed rc={rc} (3 = pool would exceed 2^31 bytes: {ln.value})")
    try:
        pool = np.ctypeslib.as_array(p, shape=(ln.value,)).copy()
    finally:
        L.svo_scene_free(p)
    return pool, st.as_dict()


def scene3_voxels(n, seed=1, amp=8, dens=CAVES_DENS, dust=0):
    """The dense voxels grid[z, y, x] of a family-1 scene (n <= 1024): what the brute-force builders start from."""
    g = np.zeros((n, n, n), dtype=np.uint8)
    rc = lib().svo_scene3_voxels(int(n), int(seed), int(amp), _d(dens, dust), g.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"svo_scene3_voxels({n}) failed rc={rc}")
    return g


def scene3_ball_counts(n, seed=1, amp=8, dens=CAVES_DENS, dust=0):
    """[[carving, solid] per level] of a family-1 / family-2 scene (the dust level, if any, comes last)"""
    c = np.zeros((5, 2), dtype=np.uint64)
    rc = lib().svo_scene3_ball_counts(int(n), int(seed), int(amp), _d(dens, dust), c.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"svo_scene3_ball_counts({n}) failed rc={rc}")
    return [[int(a), int(b)] for a, b in c]


def scene3_balls(n, seed=1, amp=8, dens=CAVES_DENS, max_balls=1 << 16):
    """int32 [count][6] = {level, cx, cy, cz, r, value (0 = carving)} of a family-1 scene's balls, coarse levels first"""
    L = lib()
    L.svo_scene3_balls.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.svo_scene3_balls.restype = ctypes.c_int
    out = np.zeros((max_balls, 6), dtype=np.int32)
    cnt = L.svo_scene3_balls(int(n), int(seed), int(amp), int(dens), out.ctypes.data, int(max_balls))
    return out[:min(cnt, max_balls)].copy()


def cave_position(n, seed=1, amp=8, dens=CAVES_DENS):
    """World position (in [1, 2]^3) of the centre of the largest cave of a family-1 scene: the carving ball of the largest radius
    among those of the coarsest level that has any -- where bench.py --camera CAVE and tests/golden/make_golden_matrix.py put a
    camera that sees the scene from the inside."""
    b = scene3_balls(n, seed, amp, dens)
    carve = b[b[:, 5] == 0]
    if carve.shape[0] == 0:
        raise RuntimeError("the scene has no carving ball")
    inner = carve[((carve[:, 1:4] - 2 * carve[:, 4:5] >= 0) & (carve[:, 1:4] + 2 * carve[:, 4:5] < n)).all(axis=1)]   # not at a world face
    if inner.shape[0]:
        carve = inner
    lvl = carve[carve[:, 0] == carve[:, 0].min()]
    c = lvl[np.argmax(lvl[:, 4])]
    return tuple(1.0 + (float(v) + 0.5) / float(n) for v in c[1:4]), int(c[4])


def build(family, n, seed=1, amp=8, dens=CAVES_DENS, dust=DUST_DENS):
    """pool, stats of scene family "terrain", "caves" or "dust" """
    if family == "terrain":
        return build_scene(n, seed, amp)
    if family == "caves":
        return build_scene3(n, seed, amp, dens)
    if family == "dust":
        return build_scene3(n, seed, amp, 0, dust)
    raise ValueError("scene family %r (terrain | caves | dust)" % (family,))


def scene_maps(n, seed=1, amp=8):
    """The procedural terrain as (height u16 [n][n], material u8 [n][n]) indexed [z][x]: the inputs of the GPU
    builder (hiplib.HipContext.build_from_heightmap), which yields the bytes build_scene() yields."""
    h = np.zeros((n, n), dtype=np.uint16)
    m = np.zeros((n, n), dtype=np.uint8)
    rc = lib().svo_scene_maps(int(n), int(seed), int(amp), h.ctypes.data, m.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"svo_scene_maps({n}) failed rc={rc}")
    return h, m


def height(n, x, z, seed=1, amp=8):
    return lib().svo_scene_height(int(n), int(seed), int(amp), int(x), int(z))


def validate_pool(pool):
    """Walk the pool; returns (rc, stats dict, max_depth). rc == 0 means consistent."""
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    st = SceneStats()
    md = ctypes.c_int()
    rc = lib().svo_pool_validate(pool.ctypes.data, pool.size, ctypes.byref(st), ctypes.byref(md))
    return rc, st.as_dict(), md.value


def embed_deep(pool, k):
    """make a (depth + k)-deep pool from a shallow one by chaining k
    all-interior nodes (child 0 interior, children 1..7 empty tag-2 leaves) above it.
    The scene then occupies [1, 1 + 2^-k]^3."""
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    if k == 0:
        return pool.copy()
    mask_chain = 0
    for n in range(1, 8):
        mask_chain |= 2 << (2 * n)
    out = bytearray()
    # chain node i sits at offset i*56 (node + its 8 children: 8*7 bytes, child 0 is the next chain node)
    # layout: [n0][c0_0..c0_7][c1_0..c1_7]... where c(i)_0 is chain node i+1
    offs = [0]
    out += bytes([1, 0, 0, 0, 0, 0, 0])
    for i in range(k):
        block = len(out)
        for n in range(8):
            out += bytes([1 if n == 0 else 0, 0, 0, 0, 0, 0, 0])
        cp = block - offs[i]
        out[offs[i] + 1:offs[i] + 5] = int(cp).to_bytes(4, "big", signed=True)
        if i < k - 1 or True:
            out[offs[i] + 5:offs[i] + 7] = int(mask_chain).to_bytes(2, "big")
        offs.append(block)
    last = offs[k]
    # last chain node adopts the shallow root's leafMask and points at its child block, appended verbatim
    root_cp = int.from_bytes(bytes(pool[1:5]), "big", signed=True)
    body = bytes(pool[root_cp:])
    cp = len(out) - last
    out[last] = int(pool[0])
    out[last + 1:last + 5] = int(cp).to_bytes(4, "big", signed=True)
    out[last + 5:last + 7] = bytes(pool[5:7])
    out += body
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()
