"""Drop-in boundary on the GPU: the JNI-typed shim called the way the JVM would call it, a pool that went
through the reference's .svo file format, the single-pixel pick, and the remaining BASELINE configs
(C2 primary-only on the persistent and staged pipelines, C5 at its stated 64 spp)."""
import ctypes
import os

import numpy as np
import pytest

import poolcache
import helpers

pytestmark = pytest.mark.gpu

PIPELINES = [int(v) for v in os.environ.get("SVO_TEST_PIPELINES", "0,1,2").split(",")]


@pytest.fixture(scope="module")
def ctx():
    from svo_raytracer_amd import hiplib
    c = helpers.DualContext()
    yield c
    c.close()


def _same(a, b):
    bad = {"rgba": int((a["rgba"] != b["rgba"]).any(axis=2).sum()),
           "depth": int((a["depth"].view(np.uint32) != b["depth"].view(np.uint32)).sum())}
    for k in ("pointer", "value", "raw_normal", "level", "iter"):
        bad[k] = int((a["hits"][k] != b["hits"][k]).sum())
    bad["t"] = int((a["hits"]["t"].view(np.uint32) != b["hits"]["t"].view(np.uint32)).sum())
    return bad


def test_jni_shim_called_with_jni_typed_arguments(ctx):
    """Every Java_src_engine_HipRenderer_n* symbol, called as the JVM calls a static native method
    (JNIEnv*, jclass, then jint / jlong / jfloat primitives and memAddress() longs): same bytes as the C ABI."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    L = hiplib.lib()
    vp, jint, jlong, jfloat = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    P = "Java_src_engine_HipRenderer_"

    def fn(name, res, *args):
        f = getattr(L, P + name)
        f.restype = res
        f.argtypes = [vp, vp] + list(args)
        return lambda *a: f(None, None, *a)

    nCreate = fn("nCreate", jlong, jint)
    nDestroy = fn("nDestroy", jint, jlong)
    nLastError = fn("nLastError", jlong, jlong)
    nPoolUpload = fn("nPoolUpload", jint, jlong, jlong, jlong)
    nPoolUpdate = fn("nPoolUpdate", jint, jlong, jlong, jlong, jlong)
    nPoolDownload = fn("nPoolDownload", jint, jlong, jlong, jlong)
    nSetCamera = fn("nSetCamera", jint, jlong, *([jfloat] * 15))
    nSetParams = fn("nSetParams", jint, jlong, *([jint] * 7))
    nResize = fn("nResize", jint, jlong, jint, jint)
    nDispatch = fn("nDispatch", jint, jlong)
    nReadColor = fn("nReadColor", jint, jlong, jlong)
    nReadDepth = fn("nReadDepth", jint, jlong, jlong)
    nReadHits = fn("nReadHits", jint, jlong, jlong)
    nReadPixel = fn("nReadPixel", jint, jlong, jint, jint, jlong, jlong, jlong)
    nReadBeam = fn("nReadBeam", jint, jlong, jlong)
    nBuildFromHeightmap = fn("nBuildFromHeightmap", jlong, jlong, jlong, jlong, jint)
    nBuildFromVoxels = fn("nBuildFromVoxels", jlong, jlong, jlong, jint)
    nSetProgressive = fn("nSetProgressive", jint, jlong, jint)
    nDerivedRefreshInfo = fn("nDerivedRefreshInfo", jlong, jlong, jlong, jlong)
    nSetPipeline = fn("nSetPipeline", jint, jlong, jint)

    pool, _ = scene.build_scene(128)
    cam = np.asarray(CAMERAS["K1"], dtype=np.float32)
    w, h = 160, 96
    ctx.set_pipeline(0)
    want = ctx.render(pool, w, h, cam, 3, 0)
    from oracle import oracle
    bad0 = _same(want, oracle.render(pool, w, h, cam, 3, 0))
    assert bad0 == {k: 0 for k in bad0}, ("C ABI vs oracle", bad0)

    assert nCreate(99) == 0                       # no such device: handle 0, like a failed GL context
    j = nCreate(0)
    assert j != 0
    try:
        assert nDispatch(j) != 0                  # no pool yet: status code, message through nLastError
        msg = ctypes.string_at(nLastError(j)).decode()
        assert "pool" in msg
        assert nPoolUpload(j, pool.ctypes.data, pool.size) == 0
        assert nSetCamera(j, *[float(v) for v in cam]) == 0
        assert nSetParams(j, 3, 0, int(pool.size), 0, 2, 0, 1) == 0
        assert nResize(j, w, h) == 0
        assert nSetPipeline(j, 1) == 0            # persistent waves on the descriptor table (HipRenderer's choice)
        assert nDispatch(j) == 0
        rgba = np.zeros((h, w, 4), dtype=np.uint8)
        depth = np.zeros((h, w), dtype=np.float32)
        hits = np.zeros((h, w), dtype=hiplib.HIT_DTYPE)
        assert nReadColor(j, rgba.ctypes.data) == 0 and nReadDepth(j, depth.ctypes.data) == 0
        assert nReadHits(j, hits.ctypes.data) == 0
        bad = _same({"rgba": rgba, "depth": depth, "hits": hits}, want)
        mm = hits["iter"] != want["hits"]["iter"]
        assert bad == {k: 0 for k in bad}, (bad, hits["iter"][mm][:12], want["hits"]["iter"][mm][:12], np.nonzero(mm)[0][:12], np.nonzero(mm)[1][:12])
        # the crosshair pick (Main.java:132-146) without the 8 MB readback
        one_c, one_d, one_h = np.zeros(4, np.uint8), np.zeros(1, np.float32), np.zeros(1, hiplib.HIT_DTYPE)
        assert nReadPixel(j, 80, 48, one_c.ctypes.data, one_d.ctypes.data, one_h.ctypes.data) == 0
        assert (one_c == rgba[48, 80]).all() and one_d.view(np.uint32)[0] == depth.view(np.uint32)[48, 80]
        assert one_h.tobytes() == hits[48, 80].tobytes()
        assert nReadPixel(j, 80, 48, 0, one_d.ctypes.data, 0) == 0      # null addresses are skipped
        assert nReadPixel(j, w, 0, 0, one_d.ctypes.data, 0) != 0        # outside the image
        # ranged update + download (Renderer.updateSSBO(start, end) / getSSBO)
        edited = pool.copy()
        ptrs = np.unique(hits["pointer"][hits["pointer"] != 0])[:64]
        edited[ptrs] = 2
        lo, hi = int(ptrs.min()), int(ptrs.max()) + 1
        assert nPoolUpdate(j, edited.ctypes.data, lo, hi) == 0
        n_states = ctypes.c_uint64(0)
        assert nDerivedRefreshInfo(j, ctypes.addressof(n_states), 0) == 1 and n_states.value >= 1   # the table followed the update
        assert nPoolUpdate(j, edited.ctypes.data, 10, 10) != 0          # start >= end: rejected like the reference
        assert nPoolUpdate(j, edited.ctypes.data, -1, 5) != 0
        back = np.zeros(pool.size, dtype=np.uint8)
        assert nPoolDownload(j, back.ctypes.data, back.size) == 0 and (back == edited).all()
        assert nDispatch(j) == 0 and nReadColor(j, rgba.ctypes.data) == 0
        want2 = ctx.render(edited, w, h, cam, 3, 0)
        assert (rgba == want2["rgba"]).all()
        # world generation and the beam image through the shim
        hm, mm = scene.scene_maps(128)
        assert nBuildFromHeightmap(j, hm.ctypes.data, mm.ctypes.data, 128) == pool.size
        assert nPoolDownload(j, back.ctypes.data, back.size) == 0 and (back == pool).all()
        assert nBuildFromHeightmap(j, hm.ctypes.data, mm.ctypes.data, 100) < 0        # not a power of two
        assert nSetParams(j, 3, 0, int(pool.size), 1, 2, 0, 1) == 0 and nDispatch(j) == 0
        beam = np.zeros(((h + 3) // 4, (w + 3) // 4), dtype=np.float32)
        assert nReadBeam(j, beam.ctypes.data) == 0
        want3 = ctx.render(pool, w, h, cam, 3, 0, use_beam=1)
        assert np.array_equal(beam.view(np.uint32), ctx.read_beam().view(np.uint32))
        assert nReadColor(j, rgba.ctypes.data) == 0 and (rgba == want3["rgba"]).all()
        small, _ = scene.build_scene(32)
        hs, ms = scene.scene_maps(32)
        yy = np.arange(32, dtype=np.int32)[None, :, None]
        grid = np.where(yy > hs.astype(np.int32)[:, None, :], 0,
                        np.where(hs.astype(np.int32)[:, None, :] - yy <= 4, ms[:, None, :], 1)).astype(np.uint8)
        assert nBuildFromVoxels(j, grid.ctypes.data, 32) == small.size
        back2 = np.zeros(small.size, dtype=np.uint8)
        assert nPoolDownload(j, back2.ctypes.data, back2.size) == 0 and (back2 == small).all()
        assert nSetProgressive(j, 1) == 0 and nSetProgressive(j, 0) == 0
    finally:
        assert nDestroy(j) == 0


@pytest.mark.parametrize("pipeline", PIPELINES)
def test_pool_loaded_from_an_svo_file_renders_like_the_oracle(ctx, pipeline, tmp_path):
    """a world saved in the reference's .svo format (4-byte big-endian memOffset + the pool,
    Octree.java:974-1012) -> Octree.readBufferFromFile -> Renderer.addSSBO -> frame, against the oracle."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hostlib
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    pool, _ = scene.build_scene(256)
    path = str(tmp_path / "level.svo")
    with open(path, "wb") as f:          # written here byte by byte as the reference's writeBufferToFile does
        f.write(int(pool.size).to_bytes(4, "big"))
        f.write(pool.tobytes())
    o = hostlib.Octree(4096)
    o.readBufferFromFile(path)
    assert o.memOffset == pool.size
    loaded = o.getByteBuffer()
    ctx.set_pipeline(pipeline)
    for mode in (0, 2):
        got = ctx.render(loaded, 240, 136, CAMERAS["K1"], 4, mode)
        ref = oracle.render(pool, 240, 136, CAMERAS["K1"], 4, mode)
        bad = _same(got, ref)
        assert bad == {k: 0 for k in bad}, (mode, bad)
    # and through the Renderer mirror's own frame loop (Main.preRun + updateEarly)
    if pipeline == PIPELINES[0]:
        c = hostlib.Camera()
        c.setPos(1.5, 1.42, 1.5)
        c.rotate(0.0, 0.3, 0.0)
        c.rotate(-0.5, 0.0, 0.0)
        o2 = hostlib.Octree(4096)
        o.writeBufferToFile(str(tmp_path / "again.svo"))
        o2.readBufferFromFile(str(tmp_path / "again.svo"))
        rgba, depth = hostlib.render_frame(o2, c, 200, 120, 2, 2)
        ref = oracle.render(pool, 200, 120, c.getUniform(), 2, 2)
        assert (rgba == ref["rgba"]).all() and (depth.view(np.uint32) == ref["depth"].view(np.uint32)).all()


def test_single_pixel_pick_equals_full_readback(ctx):
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(256)
    ctx.set_pipeline(1)
    res = ctx.render(pool, 320, 200, CAMERAS["K1"], 2, 2)
    for x, y in ((0, 0), (160, 100), (319, 199), (5, 5)):
        rgba, depth, hit = ctx.read_pixel(x, y)
        assert (rgba == res["rgba"][y, x]).all()
        assert np.float32(depth).view(np.uint32) == res["depth"].view(np.uint32)[y, x]
        assert hit.tobytes() == res["hits"][y, x].tobytes()
    with pytest.raises(hiplib.SvoError):
        ctx.read_pixel(320, 0)


# ---- BASELINE configs the first round left open -----------------------------------------------------------

@pytest.fixture(scope="module")
def pool2048():
    import svo_raytracer_amd.scene as scene
    return scene.build_scene(2048)[0]


@pytest.mark.parametrize("pipeline", [1, 2])
@pytest.mark.parametrize("mode", [1, 3])
def test_config2_2048_1080p_primary_only_vs_oracle(ctx, pool2048, pipeline, mode):
    """BASELINE config 2: 2048^3, 1920x1080, primary rays only (renderMode 1 = iteration heat map, 3 = normals: one
    cast per pixel), on the default persistent pipeline and the staged one, against the oracle on every 8th pixel."""
    from oracle import oracle
    from svo_raytracer_amd.cameras import CAMERAS
    w, h, step = 1920, 1080, 8
    ctx.set_pipeline(pipeline)
    for cam in ("K1", "K2"):
        res = ctx.render(pool2048, w, h, CAMERAS[cam], 2, mode)
        ref = oracle.render(pool2048, w, h, CAMERAS[cam], 2, mode, xstep=step, ystep=step)
        sub = (slice(0, h, step), slice(0, w, step))
        assert (ref["rgba"][sub] == res["rgba"][sub]).all()
        assert (ref["depth"].view(np.uint32)[sub] == res["depth"].view(np.uint32)[sub]).all()
        for k in ("pointer", "value", "raw_normal", "level", "iter"):
            assert (ref["hits"][k][sub] == res["hits"][k][sub]).all(), k
        assert (ref["hits"]["t"].view(np.uint32)[sub] == res["hits"]["t"].view(np.uint32)[sub]).all()
        hp = res["hits"]["pointer"]
        assert int(hp.max()) < pool2048.size and (pool2048[hp[hp != 0]] != 0).all()


def test_config5_8192_1080p_64_samples_per_pixel():
    """BASELINE config 5 at its stated 64 spp: accumulated GI, 8192^3, 1920x1080; the oracle renders the same 64
    samples (frameNumber 2..65 per pixel, summed in sample order, svotrace.comp:668-670) on 2 040 pixels."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    pool = poolcache.pool()
    c = helpers.DualContext()
    try:
        w, h, step = 1920, 1080, 32
        c.set_pipeline(1)
        res = c.render(pool, w, h, CAMERAS["K1"], 2, 0, spp=64)
        ref = oracle.render(pool, w, h, CAMERAS["K1"], 2, 0, spp=64, xstep=step, ystep=step)
        sub = (slice(0, h, step), slice(0, w, step))
        assert ref["rgba"][sub].shape[0] * ref["rgba"][sub].shape[1] >= 500
        assert (ref["rgba"][sub] == res["rgba"][sub]).all()
        assert (ref["depth"].view(np.uint32)[sub] == res["depth"].view(np.uint32)[sub]).all()
        for k in ("pointer", "value", "raw_normal", "level", "iter"):
            assert (ref["hits"][k][sub] == res["hits"][k][sub]).all(), k
        # the accumulated image is not one of its samples
        one = c.render(None, None, None, None, 2, 0, spp=1)
        assert not np.array_equal(one["rgba"], res["rgba"])
    finally:
        c.close()


@pytest.mark.parametrize("pipeline", PIPELINES)
def test_iteration_cap_inside_the_traversal_loop(ctx, pipeline):
    """Rays that skim a flat 4096^2 plateau half a voxel above it cross thousands of empty unit cells: the walk gives up
    after 1500 iterations (svotrace.comp:263-266) -- renderMode 1 paints those pixels (0.3, 0.3, 0.6), mode 2's shadow
    term reads the count.  The plateau is built by the GPU builder; the oracle renders the downloaded pool."""
    from oracle import oracle
    n = 4096
    h = np.full((n, n), n // 2, dtype=np.uint16)
    m = np.full((n, n), 2, dtype=np.uint8)
    nbytes = ctx.build_from_heightmap(h, m)
    pool = ctx.pool_download(nbytes)
    y = 1.0 + (n // 2 + 1.5) / n                      # half a voxel above the plateau's top face
    pos = (1.0 + 0.25 / n, y, 1.5)
    cam = np.array(pos + (1.0, -0.0004, -0.3) + (1.0, 0.0004, -0.3) + (1.0, -0.0004, 0.3) + (1.0, 0.0004, 0.3), dtype=np.float32)
    ctx.set_pipeline(pipeline)
    w, hh = 96, 64
    for mode in (1, 2, 0):
        got = ctx.render(None, w, hh, cam, 2, mode)
        ref = oracle.render(pool, w, hh, cam, 2, mode)
        bad = _same(got, ref)
        assert bad == {k: 0 for k in bad}, (mode, bad)
        if mode == 1:
            capped = int((ref["hits"]["iter"] > 1500).sum())
            assert capped > 200, capped                # the cap really is reached inside the loop
            cm = ref["hits"]["iter"] > 1500
            cm[:10, :10] = False                       # the debug square overwrites the corner
            assert (ref["rgba"][cm][:, :3] == [76, 76, 153]).all()
