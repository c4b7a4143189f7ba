"""Scene family 1 ("caves": the terrain under levels of hashed balls, scene/svo_scene.c) on the CPU: the generator that
never forms the voxels emits, byte for byte, what the reference builder's rules give on the dense voxels of the same scene
-- the numpy brute force (tests/poolbuilder.py) and the restated Octree.constructInnerOctree (oracle/octree_restatement.cpp,
Octree.java:511-670; parity unpinned: no JDK).  The GPU side: tests/test_gpu_scene3.py."""
import numpy as np
import pytest

import poolbuilder
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hostlib
from oracle import octree as restated


@pytest.mark.parametrize("n,seed,dens", [(32, 1, 64), (32, 2, 256), (64, 1, 64), (64, 3, 256), (128, 1, 64), (128, 2, 200)])
def test_generator_matches_dense_brute_force_builder(n, seed, dens):
    pool, st = scene.build_scene3(n, seed, 8, dens)
    grid = scene.scene3_voxels(n, seed, 8, dens)
    ref, counts = poolbuilder.pool_from_grid(grid)
    assert pool.size == ref.size and (pool == ref).all()
    for k in ("interior", "surface_leaf", "nonsurface_leaf", "subdiv_leaf"):
        assert st[k] == counts[k]
    # the balls did something: the voxels differ from the terrain's, in both directions (carved and added)
    tg = poolbuilder.terrain_grid(n, seed, 8)
    carved, added = ((grid == 0) & (tg != 0)).any(), ((grid != 0) & (tg == 0)).any()
    assert (carved and added) if n >= 128 else (carved or added)


@pytest.mark.parametrize("n,seed,dens", [(64, 1, 64), (256, 2, 64)])
def test_restated_construct_inner_octree_gives_the_same_pool(n, seed, dens):
    grid = scene.scene3_voxels(n, seed, 8, dens)
    o = hostlib.Octree(16384)
    restated.constructInnerOctree(o, grid, int(np.log2(n)))
    got = o.getByteBuffer()
    ref, _ = scene.build_scene3(n, seed, 8, dens)
    assert got.size == ref.size and (got == ref).all()


def test_overhangs_exist():
    """3-D structure the height field cannot have: columns with air below solid."""
    g = scene.scene3_voxels(128, 1, 8, 64)
    solid = g != 0                                  # [z, y, x]
    air_below_solid = (~solid[:, :-1, :]) & solid[:, 1:, :]
    assert int(air_below_solid.sum()) > 500
    t = poolbuilder.terrain_grid(128, 1, 8) != 0
    assert not ((~t[:, :-1, :]) & t[:, 1:, :]).any()


@pytest.mark.parametrize("n,depth", [(256, 8), (1024, 10), (2048, 11)])
def test_pool_validates_and_is_deterministic(n, depth):
    pool, st = scene.build_scene3(n, 1, 8, 64)
    rc, vst, maxd = scene.validate_pool(pool)
    assert rc == 0 and maxd == depth == st["depth"]
    for k in ("interior", "surface_leaf", "nonsurface_leaf", "subdiv_leaf"):
        assert st[k] == vst[k]
    assert pool.size == 7 * st["interior"] + 3 * st["surface_leaf"] + 7 * st["subdiv_leaf"] + st["nonsurface_leaf"]
    if n <= 1024:
        pool2, _ = scene.build_scene3(n, 1, 8, 64)
        assert (pool == pool2).all()
    counts = scene.scene3_ball_counts(n, 1, 8, 64)
    assert sum(a + b for a, b in counts) > 0
    # dens 0 = no balls = the terrain's bytes
    if n == 256:
        p0, _ = scene.build_scene3(n, 1, 8, 0)
        t0, _ = scene.build_scene(n, 1, 8)
        assert p0.size == t0.size and (p0 == t0).all()


@pytest.mark.parametrize("n,seed,dens,dust", [(32, 1, 0, 64), (64, 2, 0, 128), (128, 1, 0, 64), (128, 2, 64, 64)])
def test_dust_family_matches_dense_brute_force_builder(n, seed, dens, dust):
    """family 2: floating particles (radius 1 or 2 in air cells of edge max(8, n / 256)), alone and on top of the caves' balls"""
    pool, st = scene.build_scene3(n, seed, 8, dens, dust)
    grid = scene.scene3_voxels(n, seed, 8, dens, dust)
    ref, counts = poolbuilder.pool_from_grid(grid)
    assert pool.size == ref.size and (pool == ref).all()
    base = scene.scene3_voxels(n, seed, 8, dens, 0) if dens else poolbuilder.terrain_grid(n, seed, 8)
    added = (grid != 0) & (base == 0)
    assert added.sum() > 100 and not ((grid == 0) & (base != 0)).any()      # dust only adds, in the air
    if dens:
        return      # (over curved ball surfaces the six axis probes of a particle do not rule out every contact)
    # a particle floats: over the terrain alone no added voxel touches a voxel of the scene it was added to (6-neighbourhood)
    solid = base != 0
    touch = np.zeros_like(solid)
    for ax in range(3):
        for sh in (1, -1):
            r = np.roll(solid, sh, axis=ax)
            idx = [slice(None)] * 3
            idx[ax] = 0 if sh == 1 else -1
            r[tuple(idx)] = False
            touch |= r
    assert not (added & touch).any()
    assert sum(b for _, b in scene.scene3_ball_counts(n, seed, 8, dens, dust)) > 0


def test_cave_camera_sits_in_carved_air_away_from_the_world_faces():
    """cameras.cave_camera / scene.cave_position (bench.py --camera CAVE): the centre of the largest carving ball of the coarsest
    level that has one, not at a world face -- a voxel the balls emptied, with solid terrain around the cave"""
    from svo_raytracer_amd.cameras import cave_camera
    n = 256
    pos, r = scene.cave_position(n, 1, 8, 64)
    g = scene.scene3_voxels(n, 1, 8, 64)
    t = poolbuilder.terrain_grid(n, 1, 8)
    x, y, z = (int((p - 1.0) * n) for p in pos)
    assert g[z, y, x] == 0 and t[z, y, x] != 0                      # carved out of what was solid
    assert all(2 * r <= c < n - 2 * r for c in (x, y, z))
    cam = cave_camera(n, 1, 8, 64)
    assert cam.shape == (15,) and np.allclose(cam[:3], pos)
    assert (cave_camera(n, 1, 8, 64) == cam).all()
    # the scene's balls, as listed: every ball inside its cell, carving ones where the coarser scene is solid
    b = scene.scene3_balls(n, 1, 8, 64)
    assert len(b) == sum(a + s for a, s in scene.scene3_ball_counts(n, 1, 8, 64))
    for lvl, cx, cy, cz, rr, val in b[:200]:
        C = n >> (2 + 2 * lvl)
        for c in (cx, cy, cz):
            assert (c - rr) // C == (c + rr) // C == c // C
