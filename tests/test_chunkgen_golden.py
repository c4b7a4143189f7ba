"""The world builder's first stage, pinned by the reference's own shader: tests/golden/chunkgen_golden.npz holds dense chunks of
voxels that src/shaders/chunkgen-heightmap.comp made under llvmpipe (make_golden_chunkgen.py) from seeded 16-bit height maps
and material maps -- the stage Octree.constructCompleteOctree runs per chunk before constructInnerOctree (Octree.java:274-287).

Every statement of that voxel rule in this repository against it: the numpy one the tests build dense grids with
(tests/test_gpu_builder.py::dense_grid, tests/poolbuilder.py), the CPU generator's (scene/svo_scene.c::voxel, through the pool it
builds), and on the GPU build::voxel_at (csrc/svo_build.hip.h) through svo_build_from_heightmap16 -- whose pool must be
the one svo_build_from_voxels makes from the shader's voxels.  What remains unpinned of SURVEY 8(f)3 is constructInnerOctree
itself (no JDK)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def z():
    return np.load(os.path.join(HERE, "golden", "chunkgen_golden.npz"))


def chunk_of(dense_rule, raw, mat, c, ox, oy, oz):
    """a chunk cut out of the rule applied to the whole column range (heights in voxels = raw >> 5)"""
    h = (raw.astype(np.int64) >> 5)
    ys = np.arange(oy, oy + c, dtype=np.int64)[None, :, None]
    hh = h[oz:oz + c, ox:ox + c][:, None, :]
    mm = mat[oz:oz + c, ox:ox + c][:, None, :]
    return dense_rule(ys, hh, mm).astype(np.uint8)


def test_fixture_covers_the_scale_and_the_signed_material_image(z):
    assert list(z["index"]) == ["A_world", "A_part", "B_low", "B_mid", "B_high", "B_edge", "C_world"]
    assert int(z["B/raw"].max()) == 65535 and int(z["B/raw"].min()) == 0
    assert (z["B/mat"] >= 128).any() and (z["B/mat"] == 0).any()
    assert (z["B_high/voxels"] >= 128).any()          # bytes of the r8i image that are negative in the shader survive
    assert z["C_world/voxels"].shape == (128, 128, 128)


def test_numpy_voxel_rule_equals_reference_shader(z):
    """dense_grid's rule (the one tests/test_gpu_builder.py feeds the restated constructInnerOctree with)"""
    def rule(y, h, m):
        return np.where(y > h, 0, np.where(h - y <= 4, m, 1))
    for name in z["index"]:
        c, ox, oy, oz = (int(v) for v in z[name + "/meta"])
        t = name[0]
        got = chunk_of(rule, z[t + "/raw"], z[t + "/mat"], c, ox, oy, oz)
        assert np.array_equal(got, z[name + "/voxels"]), name


def test_height_scaling_is_a_shift_by_five(z):
    """int(r / 65536.0 * 2048) for every 16-bit r, as the shader evaluated it: column tops inside the golden chunks"""
    raw = z["B/raw"]
    for name in ("B_low", "B_mid", "B_high", "B_edge"):
        c, ox, oy, oz = (int(v) for v in z[name + "/meta"])
        v = z[name + "/voxels"]
        solid = v != 0
        mat0 = z["B/mat"][oz:oz + c, ox:ox + c] == 0     # material 0 leaves holes in the top five layers
        top = np.where(solid.any(axis=1), oy + c - 1 - np.argmax(solid[:, ::-1, :], axis=1), -1)   # [z][x]
        h = raw[oz:oz + c, ox:ox + c].astype(np.int64) >> 5
        inside = (h >= oy) & (h < oy + c) & ~mat0
        assert inside.sum() > (500 if name in ("B_mid", "B_high") else 20)
        assert np.array_equal(top[inside], h[inside]), name


def test_cpu_builders_agree_on_the_shader_s_voxels(z):
    """the restated constructInnerOctree and the numpy brute-force builder, both fed with the reference shader's own voxels
    (a 64^3 chunk of terrain; two 64^3 chunks of random columns with every material byte)"""
    from svo_raytracer_amd import hostlib
    from oracle import octree as restated
    from poolbuilder import pool_from_grid
    for name in ("A_part", "B_mid", "B_high"):
        grid = np.ascontiguousarray(z[name + "/voxels"])
        o = hostlib.Octree(max(4096, grid.size // 16))
        restated.constructInnerOctree(o, grid, 6)
        b, _ = pool_from_grid(grid)
        assert np.array_equal(o.getByteBuffer(), np.frombuffer(bytes(b), dtype=np.uint8)), name


@pytest.mark.gpu
@pytest.mark.parametrize("world", ["A", "C"])
def test_gpu_builder_from_raw_maps_equals_gpu_builder_from_the_shader_s_voxels(z, world):
    from svo_raytracer_amd import hiplib
    ctx = hiplib.HipContext(0)
    try:
        nb = ctx.build_from_voxels(z[world + "_world/voxels"])
        want = ctx.pool_download(nb)
        nb2 = ctx.build_from_heightmap16(z[world + "/raw"], z[world + "/mat"])
        got = ctx.pool_download(nb2)
        assert nb == nb2 and np.array_equal(got, want)
        # and the restated constructInnerOctree over the shader's voxels
        from svo_raytracer_amd import hostlib
        from oracle import octree as restated
        o = hostlib.Octree(max(4096, 128 ** 3 // 64))
        restated.constructInnerOctree(o, z[world + "_world/voxels"], 7)
        assert np.array_equal(o.getByteBuffer(), got)
    finally:
        ctx.close()
