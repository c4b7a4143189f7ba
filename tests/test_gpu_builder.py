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
    """chunkgen-heightmap.comp:16-28 over the whole world: grid[z, y, x]"""
    n = height.shape[0]
    y = np.arange(n, dtype=np.int32)[None, :, None]
    h = height.astype(np.int32)[:, None, :]
    m = material[:, None, :]
    return np.where(y > h, 0, np.where(h - y <= 4, m, 1)).astype(np.uint8)


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
