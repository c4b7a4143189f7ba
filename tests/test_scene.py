"""Scene generator (reference pool layout) host-side checks."""
import numpy as np
import pytest

import poolbuilder
import svo_raytracer_amd.scene as scene


@pytest.mark.parametrize("n", [16, 32, 64])
def test_generator_matches_dense_brute_force_builder(n):
    """The pyramid-driven C generator must emit byte-for-byte what a brute-force
    restatement of the reference builder's rules emits from the dense voxel grid."""
    pool, st = scene.build_scene(n)
    ref, counts = poolbuilder.pool_from_grid(poolbuilder.terrain_grid(n))
    assert pool.size == ref.size
    assert (pool == ref).all()
    for k in ("interior", "surface_leaf", "nonsurface_leaf", "subdiv_leaf"):
        assert st[k] == counts[k]


@pytest.mark.parametrize("n,depth", [(64, 6), (256, 8), (1024, 10)])
def test_pool_validates_and_is_deterministic(n, depth):
    pool, st = scene.build_scene(n)
    rc, vst, maxd = scene.validate_pool(pool)
    assert rc == 0 and maxd == depth == st["depth"]
    for k in ("interior", "surface_leaf", "nonsurface_leaf", "subdiv_leaf"):
        assert st[k] == vst[k]
    # record sizes add up: 7/3/7/1 bytes per interior / surface / subdividable / non-surface node
    assert pool.size == 7 * st["interior"] + 3 * st["surface_leaf"] + 7 * st["subdiv_leaf"] + st["nonsurface_leaf"]
    pool2, _ = scene.build_scene(n)
    assert (pool == pool2).all()
    # root: interior, value 1, children right behind it
    assert pool[0] == 1 and int.from_bytes(bytes(pool[1:5]), "big", signed=True) == 7


def test_multi_chunk_world_has_reference_prefix():
    """N > 1024: root + all-interior levels down to 1024^3 chunks come first (Octree.java:481-502)."""
    pool, st = scene.build_scene(2048)
    rc, _, maxd = scene.validate_pool(pool)
    assert rc == 0 and maxd == 11
    # root + 8 chunk nodes, all interior value 1 with leafMask 0
    for i in range(9):
        rec = pool[i * 7:(i + 1) * 7]
        assert rec[0] == 1 and rec[5] == 0 and rec[6] == 0
    assert pool.size < 2**31


def test_embed_deep_keeps_pool_consistent():
    pool, st = scene.build_scene(64)
    deep = scene.embed_deep(pool, 5)
    rc, vst, maxd = scene.validate_pool(deep)
    assert rc == 0 and maxd == 6 + 5
    assert vst["surface_leaf"] == st["surface_leaf"]


def test_bad_sizes_rejected():
    with pytest.raises(RuntimeError):
        scene.build_scene(100)
