"""the GPU world builder (svo_build_from_heightmap, csrc/svo_build.hip.h) -- the replacement of
Octree.constructCompleteOctree's chunk generation + constructInnerOctree + splice (Octree.java:192-353, 511-670).
Byte-for-byte against (a) the restated constructInnerOctree over the dense voxel grid that the reference's voxel
rule (chunkgen-heightmap.comp:16-28) yields from the same maps (oracle/octree_restatement.cpp; parity unpinned: no
JDK) and (b) the CPU scene generator, which applies the same rules without a grid, at the BASELINE sizes."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from svo_raytracer_amd import hiplib
    c = hiplib.HipContext(0)
    yield c
    c.close()


def dense_grid(height, material):
    """chunkgen-heightmap.comp:16-28 over the whole world: grid[z, y, x] (slab by slab: 1 GiB at 1024^3, no big temporaries)"""
    n = height.shape[0]
    g = np.empty((n, n, n), dtype=np.uint8)
    y = np.arange(n, dtype=np.int32)[:, None]
    for z in range(n):
        h = height[z].astype(np.int32)[None, :]
        g[z] = np.where(y > h, 0, np.where(h - y <= 4, material[z][None, :], 1))
    return g


def restated_pool(height, material):
    from svo_raytracer_amd import hostlib
    from oracle import octree as restated
    n = height.shape[0]
    o = hostlib.Octree(max(4096, n * n * n // 256))
    restated.constructInnerOctree(o, dense_grid(height, material), int(np.log2(n)))
    return o.getByteBuffer()


@pytest.mark.parametrize("n", [8, 16, 32, 64, 128, 256])
def test_procedural_terrain_equals_restatement_and_generator(ctx, n):
    import svo_raytracer_amd.scene as scene
    h, m = scene.scene_maps(n)
    nbytes = ctx.build_from_heightmap(h, m)
    got = ctx.pool_download(nbytes)
    ref, _ = scene.build_scene(n)
    assert got.size == ref.size and (got == ref).all()
    assert (got == restated_pool(h, m)).all()


@pytest.mark.parametrize("seed", range(6))
def test_arbitrary_maps_equal_restatement(ctx, seed):
    """Maps the generator never produces: cliffs, one-column spikes and pits, plateaus at 0 and n - 1, materials 1..9
    changing per column (so small cubes are heterogeneous in material, not only in occupancy)."""
    rng = np.random.default_rng(seed)
    n = (32, 64, 128)[seed % 3]
    base = rng.integers(0, n, size=(n // 8, n // 8))
    h = np.kron(base, np.ones((8, 8), dtype=np.int64))
    h += rng.integers(-2, 3, size=(n, n))
    spikes = rng.random((n, n)) < 0.02
    h[spikes] = rng.integers(0, n, size=int(spikes.sum()))
    if seed % 2:
        h[: n // 4] = n - 1
        h[:, : n // 8] = 0
    h = np.clip(h, 0, n - 1).astype(np.uint16)
    m = rng.integers(1, 10, size=(n, n)).astype(np.uint8)
    if seed >= 3:
        m = np.kron(rng.integers(1, 4, size=(n // 4, n // 4)), np.ones((4, 4), dtype=np.int64)).astype(np.uint8)
    nbytes = ctx.build_from_heightmap(h, m)
    got = ctx.pool_download(nbytes)
    ref = restated_pool(h, m)
    assert got.size == ref.size and (got == ref).all()
    import svo_raytracer_amd.scene as scene
    assert scene.validate_pool(got)[0] == 0


@pytest.mark.parametrize("n", [512, 1024, 2048])
def test_chunked_worlds_equal_generator(ctx, n):
    """512^3 = one task; 1024^3 = one chunk of eight tasks; 2048^3 = a level of chunk nodes above them
    (Octree.fillEmptyChildren, :481-502; splice, :317-343)."""
    import svo_raytracer_amd.scene as scene
    h, m = scene.scene_maps(n)
    nbytes = ctx.build_from_heightmap(h, m)
    got = ctx.pool_download(nbytes)
    ref, _ = scene.build_scene(n)
    assert got.size == ref.size and (got == ref).all()


def test_bench_world_8192_equals_generator_and_renders(ctx):
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    t0 = time.time()
    h, m = scene.scene_maps(8192)
    t_maps = time.time() - t0
    t0 = time.time()
    nbytes = ctx.build_from_heightmap(h, m)
    t_build = time.time() - t0
    got = ctx.pool_download(nbytes)
    t0 = time.time()
    ref, st = scene.build_scene(8192)
    t_cpu = time.time() - t0
    print("maps %.2f s, GPU build %.3f s (incl. upload of the maps), CPU generator %.2f s" % (t_maps, t_build, t_cpu))
    assert st["depth"] == 13 and got.size == ref.size and (got == ref).all()
    # the pool is live in the context: a frame straight from the built pool
    ctx.set_pipeline(1)
    res = ctx.render(None, 640, 360, CAMERAS["K1"], 2, 0)
    want = oracle.render(ref, 640, 360, CAMERAS["K1"], 2, 0, xstep=8, ystep=8)
    sub = (slice(0, 360, 8), slice(0, 640, 8))
    assert (want["rgba"][sub] == res["rgba"][sub]).all()
    assert (want["hits"]["pointer"][sub] == res["hits"]["pointer"][sub]).all()


def test_builder_rejects_bad_input(ctx):
    from svo_raytracer_amd import hiplib
    h = np.zeros((24, 24), dtype=np.uint16)
    m = np.ones((24, 24), dtype=np.uint8)
    with pytest.raises(hiplib.SvoError):
        ctx.build_from_heightmap(h, m)            # not a power of two
    h = np.zeros((32, 32), dtype=np.uint16)
    m = np.ones((32, 32), dtype=np.uint8)
    m[3, 4] = 0
    with pytest.raises(hiplib.SvoError):
        ctx.build_from_heightmap(h, m)            # material 0 = the empty voxel


def test_builder_reports_too_large_and_keeps_the_pool(ctx):
    """A dense random chunk is gigabytes of records (child pointers are signed 32-bit, Octree.java:162-168): 1024^3 of 60 %
    noise is ~4.5 GB, which wraps a 32-bit byte sum back into range -- the builder must say SVO_E_TOOLARGE (-5), write
    nothing out of bounds, and leave the context's previous pool in place."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    small, _ = scene.build_scene(64)
    ctx.set_pipeline(1)
    before = ctx.render(small, 96, 64, CAMERAS["K1"], 2, 0)
    rng = np.random.default_rng(7)
    grid = (rng.integers(0, 5, size=(1024, 1024, 1024), dtype=np.uint8) < 3).astype(np.uint8)   # 60 % solid
    with pytest.raises(hiplib.SvoError) as e:
        ctx.build_from_voxels(grid)
    assert e.value.code == -5
    del grid
    assert (ctx.pool_download(small.size) == small).all()
    after = ctx.render(None, 96, 64, CAMERAS["K1"], 2, 0)
    assert (after["rgba"] == before["rgba"]).all() and after["hits"].tobytes() == before["hits"].tobytes()


# ---- dense voxel chunks: Octree.constructInnerOctree's own input ----------------------------------------------------

def _restated_from_grid(grid):
    from svo_raytracer_amd import hostlib
    from oracle import octree as restated
    n = grid.shape[0]
    o = hostlib.Octree(max(4096, n * n * n // 128))
    restated.constructInnerOctree(o, grid, int(np.log2(n)))
    return o.getByteBuffer()


def _cave_grid(n, seed):
    """Things no height map produces: overhangs, floating islands, tunnels, hollow shells, several materials, solid and
    empty cubes of every size at every alignment."""
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    g = np.zeros((n, n, n), dtype=np.uint8)
    g[y < n // 3] = 1                                                   # ground slab
    for _ in range(6):                                                  # spheres: islands (solid) and caves (carved)
        c = rng.integers(0, n, size=3)
        r = int(rng.integers(max(2, n // 16), max(3, n // 4)))
        inside = (x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2 <= r * r
        g[inside] = int(rng.integers(0, 4))
    a = int(rng.integers(0, n // 2))
    g[a:a + n // 4, :, a:a + 2] = 3                                     # a thin wall
    g[:, n // 2:n // 2 + 1, :] = np.where(rng.random((n, 1, n)) < 0.3, 2, g[:, n // 2:n // 2 + 1, :])   # dust layer
    b = n // 8 * 3
    g[b:b + n // 8, b:b + n // 8, b:b + n // 8] = 2                      # an aligned solid cube of one material
    g[0:n // 4, n - n // 4:n, 0:n // 4] = 0                             # an aligned empty cube
    return g


@pytest.mark.parametrize("n,seed", [(2, 0), (4, 1), (8, 2), (16, 3), (32, 4), (64, 5), (64, 6), (128, 7)])
def test_voxel_chunks_equal_restated_constructInnerOctree(ctx, n, seed):
    import svo_raytracer_amd.scene as scene
    g = _cave_grid(n, seed) if n >= 8 else np.random.default_rng(seed).integers(0, 3, size=(n, n, n)).astype(np.uint8)
    nbytes = ctx.build_from_voxels(g)
    got = ctx.pool_download(nbytes)
    ref = _restated_from_grid(g)
    assert got.size == ref.size and (got == ref).all()
    assert scene.validate_pool(got)[0] == 0


@pytest.mark.parametrize("n", [64, 256, 512])
def test_voxel_chunk_of_the_terrain_equals_the_heightmap_builder(ctx, n):
    """The same world through both GPU builders (and the CPU generator): dense voxels of the maps vs the maps."""
    import svo_raytracer_amd.scene as scene
    h, m = scene.scene_maps(n)
    nb = ctx.build_from_voxels(dense_grid(h, m))
    a = ctx.pool_download(nb)
    ref, _ = scene.build_scene(n)
    assert a.size == ref.size and (a == ref).all()


def test_full_chunk_1024_from_voxels(ctx):
    """A whole 1024^3 chunk (1 GiB of voxels, the reference's unit of world generation): eight forced 512^3 tasks under
    the chunk node, neighbours outside the chunk ignored -- the bytes of the CPU generator and of the heightmap builder."""
    import svo_raytracer_amd.scene as scene
    n = 1024
    h, m = scene.scene_maps(n)
    t0 = time.time()
    g = dense_grid(h, m)
    t_grid = time.time() - t0
    t0 = time.time()
    nb = ctx.build_from_voxels(g)
    t_build = time.time() - t0
    del g
    got = ctx.pool_download(nb)
    ref, _ = scene.build_scene(n)
    print("1024^3 voxels: numpy grid %.1f s, GPU build %.2f s (incl. 1 GiB upload), pool %d bytes" % (t_grid, t_build, nb))
    assert got.size == ref.size and (got == ref).all()
