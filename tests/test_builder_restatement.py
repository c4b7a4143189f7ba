"""SURVEY 8(f) row 3 on the CPU: the three independent statements of the reference's builder rules agree byte for
byte -- the restated Octree.constructInnerOctree over a dense voxel chunk (oracle/octree_restatement.cpp,
Octree.java:511-670; parity unpinned: no JDK), the procedural generator that never materialises the grid
(scene/svo_scene.c) and the numpy brute force (tests/poolbuilder.py).  The HIP builder is checked against the same
bytes in tests/test_gpu_builder.py."""
import numpy as np
import pytest

from svo_raytracer_amd import hostlib
import svo_raytracer_amd.scene as scene
from oracle import octree as restated


@pytest.mark.parametrize("n", [16, 32, 64])
def test_construct_inner_octree_matches_generator_and_brute_force(n):
    import poolbuilder
    grid = poolbuilder.terrain_grid(n)
    o = hostlib.Octree(4096)
    depth = int(np.log2(n))
    restated.constructInnerOctree(o, grid, depth)
    got = o.getByteBuffer()
    ref, _ = scene.build_scene(n)
    assert got.size == ref.size and (got == ref).all()
    brute, _ = poolbuilder.pool_from_grid(grid)
    assert (got == brute).all()
