"""Edge cases of the boundary, every pipeline against the oracle: degenerate images (1x1, one row, one column, sizes
just around the 8-pixel tile and the 4-pixel beam block), the smallest legal pools, cameras with zero / infinite /
NaN components and zero-length corner rays, the camera far outside or exactly on a face of the cube -- with and without
the beam pre-pass, one frame per dispatch and batched."""
import os

import numpy as np
import pytest
import helpers

pytestmark = pytest.mark.gpu

PIPELINES = [int(v) for v in os.environ.get("SVO_TEST_PIPELINES", "0,1,2").split(",")]


@pytest.fixture(scope="module")
def ctx():
    from svo_raytracer_amd import hiplib
    c = helpers.DualContext()
    yield c
    c.close()


def _check(ctx, pool, w, h, cam, frame, mode, beam, tag):
    from oracle import oracle
    got = ctx.render(pool, w, h, cam, frame, mode, use_beam=beam)
    ref = oracle.render(pool, w, h, cam, frame, mode, use_beam=bool(beam))
    assert (got["rgba"] == ref["rgba"]).all(), tag
    assert (got["depth"].view(np.uint32) == ref["depth"].view(np.uint32)).all(), tag
    for k in ("pointer", "value", "raw_normal", "level", "iter"):
        assert (got["hits"][k] == ref["hits"][k]).all(), tag + (k,)
    assert (got["hits"]["t"].view(np.uint32) == ref["hits"]["t"].view(np.uint32)).all(), tag


@pytest.mark.parametrize("pipeline", PIPELINES)
def test_degenerate_image_sizes(ctx, pipeline):
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(64)
    ctx.set_pipeline(pipeline)
    for w, h in ((1, 1), (1, 37), (53, 1), (3, 3), (4, 4), (5, 9), (7, 8), (8, 7), (9, 9), (15, 17), (64, 3), (2, 130)):
        for beam in (0, 1):
            _check(ctx, pool, w, h, CAMERAS["K1"], 2, 0, beam, (pipeline, w, h, beam))
            _check(ctx, pool, w, h, CAMERAS["K0"], 3, 2, beam, (pipeline, w, h, beam, "m2"))


@pytest.mark.parametrize("pipeline", PIPELINES)
def test_smallest_pools(ctx, pipeline):
    from svo_raytracer_amd.cameras import CAMERAS
    ctx.set_pipeline(pipeline)
    root_only = np.array([1, 0, 0, 0, 0, 0, 0], dtype=np.uint8)                       # a root without children
    self_loop = np.array([1, 0, 0, 0, 0, 0, 0, 0], dtype=np.uint8)                    # cp = 0: children = the root itself
    one_level = np.array([1, 0, 0, 0, 7, 0xff, 0xff] + [1, 0, 2, 0, 3, 0, 0, 5], dtype=np.uint8)   # 8 one-byte leaves
    surf = [1, 0, 0, 0, 7, 0x55, 0x55]
    for i in range(8):
        surf += [1 + i % 3, (455 + 20 * i) & 0xff, (455 + 20 * i) >> 8]                # 8 surface leaves, assorted normals
    for pool in (root_only, self_loop, one_level, np.array(surf, dtype=np.uint8)):
        for cam in ("K0", "K1", "K2"):
            for mode in (0, 1, 2, 3):
                for beam in (0, 1):
                    _check(ctx, pool, 40, 24, CAMERAS[cam], 2, mode, beam, (pipeline, pool.size, cam, mode, beam))


@pytest.mark.parametrize("pipeline", PIPELINES)
def test_hostile_cameras(ctx, pipeline):
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(128)
    ctx.set_pipeline(pipeline)
    base = np.array(CAMERAS["K1"], dtype=np.float32)
    cams = []
    for idx, val in ((0, np.nan), (1, np.inf), (2, -np.inf), (4, np.nan), (7, np.inf), (9, 0.0), (13, -0.0), (5, 1e30), (3, 1e-30)):
        c = base.copy()
        c[idx] = val
        cams.append(c)
    z = base.copy(); z[3:] = 0.0; cams.append(z)                                      # all corner rays zero: normalize(0) = NaN
    n = base.copy(); n[:] = np.nan; cams.append(n)
    a = base.copy(); a[3:6] = a[6:9] = a[9:12] = a[12:15] = (0.0, -1.0, 0.0); cams.append(a)   # one direction, axis-aligned
    f = base.copy(); f[:3] = (1.0, 1.5, 1.5); cams.append(f)                           # exactly on a face of the cube
    o = base.copy(); o[:3] = (40.0, 30.0, -25.0); cams.append(o)                       # far outside
    i = base.copy(); i[:3] = (1.5, 1.01, 1.5); cams.append(i)                          # inside the terrain (solid voxels)
    for k, cam in enumerate(cams):
        for mode in (0, 2, 3):
            for beam in (0, 1):
                _check(ctx, pool, 48, 32, cam, 2 + k, mode, beam, (pipeline, k, mode, beam))


def test_batches_of_hostile_frames(ctx):
    """The same hostile cameras through batched dispatches (persistent pipeline: one launch per batch)."""
    import torch
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    pool, _ = scene.build_scene(128)
    w, h, nb = 48, 32, 5
    ctx.set_pipeline(1)
    ctx.pool_upload(pool)
    ctx.resize(w, h)
    col = torch.zeros((nb, h, w), dtype=torch.int32, device="cuda")
    dep = torch.zeros((nb, h, w), dtype=torch.float32, device="cuda")
    base = np.array(CAMERAS["K2"], dtype=np.float32)
    try:
        for idx, val in ((0, np.nan), (4, np.inf), (9, 0.0), (14, np.nan)):
            cam = base.copy()
            cam[idx] = val
            ctx.set_camera(cam)
            ctx.bind_outputs(col.data_ptr(), dep.data_ptr(), None)
            ctx.set_batch(nb, w * h)
            for beam in (0, 1):
                ctx.set_params(7, 0, 0, beam, 3, 0b100, 1)
                ctx.dispatch()
                for k in range(nb):
                    ref = oracle.render(pool, w, h, cam, 7 + k, 0, bounces=3, mirror_mask=0b100, use_beam=bool(beam))
                    assert np.array_equal(col[k].cpu().numpy().view(np.uint8).reshape(h, w, 4), ref["rgba"]), (idx, beam, k)
                    assert np.array_equal(dep[k].cpu().numpy().view(np.uint32), ref["depth"].view(np.uint32)), (idx, beam, k)
    finally:
        ctx.set_batch(1, 0)
        ctx.bind_outputs(None, None, None)


def test_largest_pool_and_largest_frame(ctx):
    """Maximum sizes: a pool one byte short of the reference's limit (a Java byte[] / GL int offsets: 2^31 - 1 bytes,
    Octree.java:31-36) whose nodes all live in its last megabytes -- every child pointer of the walk is an offset close
    to 2^31 -- rendered into an 8K frame (7680x4320: 518 400 tiles, 33 M pixels); one byte more is refused."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    small, _ = scene.build_scene(256)
    total = (1 << 31) - 1
    off = total - small.size
    big = np.zeros(total, dtype=np.uint8)
    big[off:] = small
    cp = int.from_bytes(bytes(small[1:5]), "big", signed=True)
    big[0] = small[0]
    big[1:5] = np.frombuffer(int(off + cp).to_bytes(4, "big", signed=True), dtype=np.uint8)   # the root's children: at the far end
    big[5:7] = small[5:7]
    assert scene.validate_pool(big)[0] == 0
    w, h = 7680, 4320
    sub = (slice(0, h, 89), slice(0, w, 97))
    refs = {beam: oracle.render(big, w, h, CAMERAS["K1"], 5, 0, xstep=97, ystep=89, use_beam=bool(beam)) for beam in (0, 1)}
    ref_small = oracle.render(small, w, h, CAMERAS["K1"], 5, 0, xstep=97, ystep=89)
    first = True
    for pipeline in PIPELINES:
        ctx.set_pipeline(pipeline)
        for beam in (0, 1):
            got = ctx.render(big if first else None, w, h, CAMERAS["K1"], 5, 0, use_beam=beam)
            first = False
            ref = refs[beam]
            assert (got["rgba"][sub] == ref["rgba"][sub]).all(), (pipeline, beam)
            assert (got["depth"][sub].view(np.uint32) == ref["depth"][sub].view(np.uint32)).all(), (pipeline, beam)
            for k in ("pointer", "value", "raw_normal", "level", "iter"):
                assert (got["hits"][k][sub] == ref["hits"][k][sub]).all(), (pipeline, beam, k)
            assert int(got["hits"]["pointer"].max()) > (1 << 31) - small.size
            # the same frame from the small pool: only the pointers move
            assert (got["rgba"][sub] == ref_small["rgba"][sub]).all()
            assert (got["hits"]["pointer"][sub].astype(np.int64) - ref_small["hits"]["pointer"][sub].astype(np.int64) ==
                    np.where(ref_small["hits"]["pointer"][sub] != 0, off, 0)).all()
    with pytest.raises(hiplib.SvoError):
        ctx.pool_upload(np.zeros(1 << 31, dtype=np.uint8))
    ctx.resize(64, 64)   # give the 8K images back
