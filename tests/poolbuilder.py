"""Dense-grid -> SVO byte pool builder for tests (numpy, small N only).

An independent, brute-force statement of the reference builder's rules
(/root/reference/src/engine/Octree.java:511-670) used to
  * cross-check the C scene generator (which never materialises a grid), and
  * build hand-crafted edge-case scenes (dust lattices, stale tag-2 masks, ...).

grid[z, y, x] is a uint8 voxel value (0 = empty).
"""
import numpy as np

NODE, LEAF, NSLEAF = 7, 3, 1
T_INTERIOR, T_SURFACE, T_SUBDIV, T_NONSURFACE = 0, 1, 2, 3


class PoolBuilder:
    def __init__(self, grid, chunk=1024, task_size=512):
        self.g = np.ascontiguousarray(grid, dtype=np.uint8)
        self.n = self.g.shape[0]
        assert self.g.shape == (self.n, self.n, self.n)
        self.chunk = min(chunk, self.n)
        self.out = bytearray()
        self.counts = {"interior": 0, "surface_leaf": 0, "nonsurface_leaf": 0, "subdiv_leaf": 0}

    # --- encoders (Octree.java:119-176)
    def node7(self, val):
        p = len(self.out)
        self.out += bytes([val, 0, 0, 0, 0, 0, 0])
        return p

    def surface(self, val, normal):
        p = len(self.out)
        self.out += bytes([val, normal & 0xFF, normal >> 8])
        return p

    def nsleaf(self, val):
        p = len(self.out)
        self.out += bytes([val])
        return p

    def set_cp(self, parent, child):
        self.out[parent + 1:parent + 5] = int(child - parent).to_bytes(4, "big", signed=True)

    def set_mask(self, parent, mask):
        self.out[parent + 5:parent + 7] = int(mask).to_bytes(2, "big")

    def in_chunk(self, gcoord, c):
        o = (c // self.chunk) * self.chunk
        return o <= gcoord < o + self.chunk and 0 <= gcoord < self.n

    # --- Octree.java:620-649
    def surface_normal(self, x, y, z):
        exposed, nx, ny, nz = False, 0, 0, 0
        for i in (x - 1, x, x + 1):
            if not self.in_chunk(i, x):
                continue
            for j in (y - 1, y, y + 1):
                if not self.in_chunk(j, y):
                    continue
                for k in (z - 1, z, z + 1):
                    if not self.in_chunk(k, z):
                        continue
                    if self.g[k, j, i] == 0:
                        exposed = True
                        nx += i - x
                        ny += j - y
                        nz += k - z
        t = lambda v: int(v / 2) + 5  # noqa: E731  (truncate toward zero like Java)
        return exposed, t(nx) + 10 * t(ny) + 100 * t(nz)

    # --- Octree.java:651-670
    def big_exposed(self, x, y, z, cs):
        for k in (z - 1, z + cs, z + cs + 1):
            if not self.in_chunk(k, z):
                continue
            for j in (y - 1, y + cs, y + cs + 1):
                if not self.in_chunk(j, y):
                    continue
                for i in (x - 1, x + cs, x + cs + 1):
                    if not self.in_chunk(i, x):
                        continue
                    if self.g[k, j, i] == 0:
                        return True
        return False

    # --- Octree.java:527-555
    def classify(self, x, y, z, cs):
        sub = self.g[z:z + cs, y:y + cs, x:x + cs]
        first = int(sub[0, 0, 0])
        if cs == 1:
            return True, first
        flat = sub.reshape(-1)  # z outer, y middle, x inner == the reference scan order
        diff = np.nonzero(flat != first)[0]
        if diff.size == 0:
            return True, first
        if first == 0:
            first = int(flat[diff[0]])
        return False, first

    def build_children(self, parent, px, py, pz, size):
        cs = size // 2
        if cs == 0:
            return
        offs, vals, types = [], [], []
        mask = 0
        for n in range(8):
            x, y, z = px + (n & 1) * cs, py + ((n >> 1) & 1) * cs, pz + ((n >> 2) & 1) * cs
            leaf, val = self.classify(x, y, z, cs)
            if leaf and val != 0:
                if cs == 1:
                    exposed, nrm = self.surface_normal(x, y, z)
                    if exposed:
                        offs.append(self.surface(val, nrm)); t = T_SURFACE; self.counts["surface_leaf"] += 1
                    else:
                        offs.append(self.nsleaf(val)); t = T_NONSURFACE; self.counts["nonsurface_leaf"] += 1
                elif self.big_exposed(x, y, z, cs):
                    offs.append(self.node7(val)); t = T_INTERIOR; self.counts["interior"] += 1
                else:
                    offs.append(self.node7(val)); t = T_SUBDIV; self.counts["subdiv_leaf"] += 1
            elif leaf:
                if cs == 1:
                    offs.append(self.nsleaf(val)); t = T_NONSURFACE; self.counts["nonsurface_leaf"] += 1
                else:
                    offs.append(self.node7(val)); t = T_SUBDIV; self.counts["subdiv_leaf"] += 1
            else:
                offs.append(self.node7(val)); t = T_INTERIOR; self.counts["interior"] += 1
            vals.append(val)
            types.append(t)
            mask |= t << (2 * n)
        self.set_cp(parent, offs[0])
        self.set_mask(parent, mask)
        for n in range(8):
            if vals[n] != 0 and types[n] == T_INTERIOR:
                x, y, z = px + (n & 1) * cs, py + ((n >> 1) & 1) * cs, pz + ((n >> 2) & 1) * cs
                self.build_children(offs[n], x, y, z, cs)

    def build(self):
        root = self.node7(1)
        self.counts["interior"] += 1
        assert self.n <= 512, "test builder covers single-task worlds only"
        self.build_children(root, 0, 0, 0, self.n)
        return np.frombuffer(bytes(self.out), dtype=np.uint8).copy()


def pool_from_grid(grid):
    b = PoolBuilder(grid)
    return b.build(), b.counts


def terrain_grid(n, seed=1, amp=8):
    """Dense grid of the procedural terrain (voxel rule of chunkgen-heightmap.comp:16-28)."""
    import svo_raytracer_amd.scene as scene
    L = scene.lib()
    h = np.array([[L.svo_scene_height(n, seed, amp, x, z) for x in range(n)] for z in range(n)], dtype=np.int32)
    ys = np.arange(n, dtype=np.int32)[None, :, None]
    hh = h[:, None, :]
    # band material: same hash as scene/svo_scene.c band_material()
    def mix32(a):
        a = np.uint32(a)
        a ^= a >> np.uint32(16); a = (a * np.uint32(0x7feb352d)) & np.uint32(0xFFFFFFFF)
        a ^= a >> np.uint32(15); a = (a * np.uint32(0x846ca68b)) & np.uint32(0xFFFFFFFF)
        a ^= a >> np.uint32(16)
        return a
    xs = np.arange(n, dtype=np.uint32)[None, :] >> np.uint32(5)
    zs = np.arange(n, dtype=np.uint32)[:, None] >> np.uint32(5)
    with np.errstate(over="ignore"):
        inner = mix32(zs + np.uint32(seed) * np.uint32(0x61C88647))
        band = (np.uint32(2) + (mix32((xs * np.uint32(0x9E3779B1)) ^ inner) & np.uint32(1))).astype(np.uint8)
    grid = np.zeros((n, n, n), dtype=np.uint8)
    solid = ys <= hh
    inband = (hh - ys) <= 4
    grid[solid & ~inband] = 1
    bb = np.broadcast_to(band[:, None, :], (n, n, n))
    m = solid & inband
    grid[m] = bb[m]
    return grid


def dust_grid(n=128, floor=20, cell=8, seed=7):
    """A solid floor with a lattice of isolated single voxels ('dust') above it: rays
    descend into many occupied coarse cells and miss -> very high iteration counts
    (exercises the iter > 260 penumbra branch, svotrace.comp:616-619)."""
    g = np.zeros((n, n, n), dtype=np.uint8)
    g[:, :floor + 1, :] = 1
    rng = np.random.RandomState(seed)
    for cz in range(0, n, cell):
        for cy in range(((floor + cell) // cell + 1) * cell, n, cell):
            for cx in range(0, n, cell):
                ox, oy, oz = rng.randint(1, cell - 1, size=3)
                g[cz + oz, cy + oy, cx + ox] = 2
    return g
