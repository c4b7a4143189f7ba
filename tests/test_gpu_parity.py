"""GPU parity tests proper: the HIP path, called through the C ABI (libsvohip.so),
against (a) the golden vectors the reference shader produced under llvmpipe and
(b) the CPU oracle on seeded synthetic scenes.  Bar: bit-exact everywhere
(rgba8, depth bits, hit pointer / value / raw normal / level / iteration count / t bits)."""
import numpy as np
import pytest

from helpers import compare_with_golden, golden_case, golden_cases
import helpers

pytestmark = pytest.mark.gpu

import os
PIPELINES = [int(v) for v in os.environ.get("SVO_TEST_PIPELINES", "0,1,2").split(",")]


@pytest.fixture(scope="module")
def ctx():
    from svo_raytracer_amd import hiplib
    c = helpers.DualContext()
    yield c
    c.close()


def _same(a, b):
    bad = {
        "rgba": int((a["rgba"] != b["rgba"]).any(axis=2).sum()),
        "depth": int((a["depth"].view(np.uint32) != b["depth"].view(np.uint32)).sum()),
    }
    for k in ("pointer", "value", "raw_normal", "level", "iter"):
        bad[k] = int((a["hits"][k] != b["hits"][k]).sum())
    bad["t"] = int((a["hits"]["t"].view(np.uint32) != b["hits"]["t"].view(np.uint32)).sum())
    return bad


@pytest.mark.parametrize("pipeline", PIPELINES)
@pytest.mark.parametrize("name,poolkey", golden_cases())
def test_hip_matches_reference_shader_golden(ctx, name, poolkey, pipeline):
    g = golden_case(name, poolkey)
    ctx.set_pipeline(pipeline)
    res = ctx.render(g["pool"], g["w"], g["h"], g["cam"], g["frame"], g["mode"])
    bad = compare_with_golden(res, g)
    assert bad == {k: 0 for k in bad}, bad


@pytest.mark.parametrize("pipeline", PIPELINES)
@pytest.mark.parametrize("n,w,h,mode,cam", [
    (512, 256, 256, 1, "K0"),     # BASELINE config 1: 512^3, 256x256, primary rays only
    (512, 256, 256, 2, "K1"),
    (512, 320, 200, 0, "K1"),
    (1024, 200, 120, 0, "K2"),
    (1024, 200, 120, 2, "K0"),
])
def test_hip_matches_oracle_on_synthetic_scenes(ctx, n, w, h, mode, cam, pipeline):
    import svo_raytracer_amd.scene as scene
    from oracle import oracle
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(n)
    ctx.set_pipeline(pipeline)
    res = ctx.render(pool, w, h, CAMERAS[cam], 2, mode)
    ref = oracle.render(pool, w, h, CAMERAS[cam], 2, mode)
    bad = _same(res, ref)
    assert bad == {k: 0 for k in bad}, bad


@pytest.mark.parametrize("pipeline", PIPELINES)
@pytest.mark.parametrize("bounces,mirror,spp", [(1, 0, 1), (3, 0, 1), (4, 0b0100, 1), (2, 0, 4), (3, 0b1000, 3)])
def test_extended_path_options_match_oracle(ctx, bounces, mirror, spp, pipeline):
    """bounces / mirror materials / spp: the shader's dormant features (svotrace.comp:444,
    500-504, 668-670) as the oracle defines them."""
    import svo_raytracer_amd.scene as scene
    from oracle import oracle
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(256)
    ctx.set_pipeline(pipeline)
    res = ctx.render(pool, 160, 96, CAMERAS["K1"], 5, 0, bounces=bounces, mirror_mask=mirror, spp=spp)
    ref = oracle.render(pool, 160, 96, CAMERAS["K1"], 5, 0, bounces=bounces, mirror_mask=mirror, spp=spp)
    bad = _same(res, ref)
    assert bad == {k: 0 for k in bad}, bad


def test_counters_match_oracle(ctx):
    import svo_raytracer_amd.scene as scene
    from oracle import oracle
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(256)
    ctx.set_pipeline(0)
    for mode in (0, 2):
        ctx.render(pool, 128, 96, CAMERAS["K1"], 2, mode)
        st = ctx.count_frame()
        ref = oracle.render(pool, 128, 96, CAMERAS["K1"], 2, mode)["stats"]
        for k in ("pixels", "rays", "nan_rays", "iterations", "alg_bytes", "max_iter"):
            assert st[k] == ref[k], (mode, k, st[k], ref[k])


def test_row_split_equals_full_frame(ctx):
    """Screen-tile sharding: rendering row bands separately gives the same bytes."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(256)
    ctx.set_pipeline(0)
    full = ctx.render(pool, 200, 120, CAMERAS["K1"], 2, 0)
    ctx.resize(200, 120)
    parts = np.zeros_like(full["rgba"])
    for y0, y1 in ((0, 32), (32, 64), (64, 96), (96, 120)):
        ctx.set_rows(y0, y1)
        ctx.dispatch()
        parts[y0:y1] = ctx.read_color()[y0:y1]
    ctx.set_rows(0, 120)
    assert (parts == full["rgba"]).all()


@pytest.mark.parametrize("pipeline", PIPELINES)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_interleaved_stripes_reassemble_the_frame(ctx, pipeline, world):
    """Load-balanced sharding: rank r renders tile rows r, r+N, ... packed into its band of the gather buffer."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from svo_raytracer_amd.tiles import stripe_layout, deinterleave
    pool, _ = scene.build_scene(256)
    w, h = 200, 116                      # 15 tile rows, the last one partial
    ctx.set_pipeline(pipeline)
    full = ctx.render(pool, w, h, CAMERAS["K1"], 2, 0)
    rpr = stripe_layout(h, world, 0)[4]
    import torch
    col = torch.zeros((rpr * world, w), dtype=torch.int32, device="cuda")
    dep = torch.zeros((rpr * world, w), dtype=torch.float32, device="cuda")
    ctx.resize(w, h)
    ctx.bind_outputs(col.data_ptr(), dep.data_ptr(), None)
    for r in range(world):
        first, step, n, out0, rows = stripe_layout(h, world, r)
        ctx.set_stripes(first, step, n, out0)
        ctx.dispatch()
    torch.cuda.synchronize()
    ctx.bind_outputs(None, None, None)
    ctx.set_rows(0, h)
    got = deinterleave(col.cpu().numpy().view(np.uint8).reshape(rpr * world, w, 4), world, rpr, h)
    gotd = deinterleave(dep.cpu().numpy(), world, rpr, h)
    assert (got == full["rgba"]).all()
    assert (gotd.view(np.uint32) == full["depth"].view(np.uint32)).all()


def test_pool_update_and_errors(ctx):
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from oracle import oracle
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(128)
    ctx.set_pipeline(0)
    base = ctx.render(pool, 96, 64, CAMERAS["K1"], 2, 2)
    # edit material bytes of a range of the pool and send only that range (Renderer.updateSSBO ranged form)
    edited = pool.copy()
    hit = base["hits"]
    ptrs = np.unique(hit["pointer"][hit["pointer"] != 0])[:200]
    edited[ptrs] = 3
    lo, hi = int(ptrs.min()), int(ptrs.max()) + 1
    ctx.pool_update(edited, lo, hi)
    ctx.dispatch()
    res = {"rgba": ctx.read_color(), "depth": ctx.read_depth(), "hits": ctx.read_hits()}
    ref = oracle.render(edited, 96, 64, CAMERAS["K1"], 2, 2)
    bad = _same(res, ref)
    assert bad == {k: 0 for k in bad}, bad
    assert (ctx.pool_download(pool.size) == edited).all()
    # the reference rejects start >= end (Renderer.java:137-140)
    with pytest.raises(hiplib.SvoError):
        ctx.pool_update(edited, 10, 10)


def test_full_size_frame_properties(ctx):
    """BASELINE-size frame (1920x1080, 2048^3): size-independent properties -- determinism,
    hit pointers inside the pool and landing on non-empty nodes, depth > 0 exactly on hits,
    sky pixels exactly where the oracle (subsampled) says."""
    import svo_raytracer_amd.scene as scene
    from oracle import oracle
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(2048)
    ctx.set_pipeline(0)
    a = ctx.render(pool, 1920, 1080, CAMERAS["K1"], 2, 2)
    b = ctx.render(None, None, None, None, 2, 2)
    assert (a["rgba"] == b["rgba"]).all() and (a["depth"].view(np.uint32) == b["depth"].view(np.uint32)).all()
    hits = a["hits"]
    hp = hits["pointer"]
    assert int(hp.max()) < pool.size
    assert (pool[hp[hp != 0]] != 0).all()          # a hit node is never empty
    assert (pool[hp[hp != 0]] == hits["value"][hp != 0]).all()
    assert ((a["depth"] > 0) == (hp != 0)).all()
    ref = oracle.render(pool, 1920, 1080, CAMERAS["K1"], 2, 2, xstep=16, ystep=16)
    sub = (slice(0, 1080, 16), slice(0, 1920, 16))
    assert (ref["hits"]["pointer"][sub] == hp[sub]).all()
    assert (ref["rgba"][sub] == a["rgba"][sub]).all()
    assert (ref["depth"].view(np.uint32)[sub] == a["depth"].view(np.uint32)[sub]).all()


# ---- BASELINE.json configs at full size: subsampled oracle + size-independent properties ----------

def _check_subsampled(ctx, pool, w, h, cam, frame, mode, step, **opts):
    import numpy as np
    from oracle import oracle
    res = ctx.render(pool, w, h, cam, frame, mode, **opts)
    ref = oracle.render(pool, w, h, cam, frame, mode, xstep=step, ystep=step, **opts)
    sub = (slice(0, h, step), slice(0, w, step))
    assert (ref["rgba"][sub] == res["rgba"][sub]).all()
    assert (ref["depth"].view(np.uint32)[sub] == res["depth"].view(np.uint32)[sub]).all()
    for k in ("pointer", "value", "raw_normal", "level", "iter"):
        assert (ref["hits"][k][sub] == res["hits"][k][sub]).all(), k
    hp = res["hits"]["pointer"]
    assert int(hp.max()) < pool.size and (pool[hp[hp != 0]] != 0).all()
    return res


@pytest.fixture(scope="module")
def pool8192():
    import poolcache
    pool = poolcache.pool()
    assert pool.size < 2**31
    return pool


@pytest.mark.parametrize("pipeline", PIPELINES)
def test_config3_8192_1080p_primary_plus_bounce(ctx, pool8192, pipeline):
    """BASELINE config 3 (the metric's config): 8192^3, 1920x1080, primary + 1 diffuse bounce."""
    from svo_raytracer_amd.cameras import CAMERAS
    ctx.set_pipeline(pipeline)
    a = _check_subsampled(ctx, pool8192, 1920, 1080, CAMERAS["K1"], 2, 0, 24)
    b = ctx.render(None, None, None, None, 2, 0)
    assert (a["rgba"] == b["rgba"]).all()   # idempotent: same frame, same bytes


def test_config4_8192_4k_four_bounces_with_mirror(ctx, pool8192):
    """BASELINE config 4: 3840x2160, 4 bounces incl. mirror materials (single GPU here; the
    tile split is covered by test_row_split_equals_full_frame and the gloo test)."""
    from svo_raytracer_amd.cameras import CAMERAS
    ctx.set_pipeline(1)
    _check_subsampled(ctx, pool8192, 3840, 2160, CAMERAS["K1"], 2, 0, 48, bounces=5, mirror_mask=0b1000)


def test_spp8_8192_1080p_the_library_s_own_reading_of_the_sample_loop(ctx, pool8192):
    """NOT BASELINE config 5 (that is the reference's 64-frame accumulation: tests/test_c5_progressive.py, and through the
    8-way split tests/test_gpu_group.py).  This is `spp`, the builder's reading of the shader's commented-out sample loop
    (svotrace.comp:668-670; seeds frameNumber + sample, fp32 mean): 8 samples per pixel at 8192^3 / 1080p, HIP <-> oracle only
    -- there is no reference behaviour to compare with."""
    from svo_raytracer_amd.cameras import CAMERAS
    ctx.set_pipeline(1)
    _check_subsampled(ctx, pool8192, 1920, 1080, CAMERAS["K2"], 2, 0, 40, spp=8)


def test_c_abi_error_behaviour():
    """Status codes instead of exceptions or prints (include/svo_hip.h conventions)."""
    from svo_raytracer_amd import hiplib
    c = helpers.DualContext()
    try:
        c.resize(64, 48)
        with pytest.raises(hiplib.SvoError) as e:
            c.dispatch()                       # no pool yet
        assert e.value.code == -2              # SVO_E_NOPOOL
        with pytest.raises(hiplib.SvoError):
            c.resize(0, 10)
        with pytest.raises(hiplib.SvoError):
            c.set_params(2, 0, 0, 0, bounces=0)
        with pytest.raises(hiplib.SvoError):
            c.set_pipeline(7)
        with pytest.raises(hiplib.SvoError):
            c.set_rows(4, 16)                  # y0 must be a multiple of the 8-pixel tile
        # an all-zero pool is legal: every ray misses (root value / pointers are 0)
        c.pool_upload(np.zeros(64, dtype=np.uint8))
        out = c.render(None, 64, 48, np.asarray([1.5, 1.5, 2.0, -1.6, -0.9, -1, -1.6, 0.9, -1, 1.6, -0.9, -1, 1.6, 0.9, -1],
                                                dtype=np.float32), 2, 2)
        assert (out["hits"]["pointer"] == 0).all() and (out["depth"] == 0).all()
        assert (out["rgba"][:10, :10, :3] == [255, 0, 0]).all()   # first dword 0 -> red debug square
    finally:
        c.close()


def test_empty_pool_matches_oracle(ctx):
    from oracle import oracle
    from svo_raytracer_amd.cameras import CAMERAS
    pool = np.zeros(64, dtype=np.uint8)
    for pipeline in PIPELINES:
        ctx.set_pipeline(pipeline)
        for mode in (0, 1, 2, 3):
            got = ctx.render(pool, 72, 40, CAMERAS["K0"], 2, mode)
            ref = oracle.render(pool, 72, 40, CAMERAS["K0"], 2, mode)
            assert (got["rgba"] == ref["rgba"]).all(), (pipeline, mode)
            assert (got["depth"].view(np.uint32) == ref["depth"].view(np.uint32)).all()
            assert (got["hits"]["iter"] == ref["hits"]["iter"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_asm_loop_equals_cxx_loop(ctx, mode):
    """The hand-written gfx950 traversal loop (csrc/svo_travloop.h) against hipcc's translation of the readable
    trav_step() (csrc/svo_trav.h, library built with -DSVO_ASM_LOOP=0): every output of a full-HD frame, bit for bit,
    including iteration counts -- on a regular camera, on one with NaN components and on one outside the cube."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    if not os.path.exists(hiplib.CXXLOOP_LIB_PATH):
        pytest.fail("libsvohip_cxxloop.so missing: run __graft_entry__.build()")
    other = hiplib.HipContext(0, lib_path=hiplib.CXXLOOP_LIB_PATH)
    try:
        pool, _ = scene.build_scene(2048)
        nan_cam = np.array(CAMERAS["K1"], dtype=np.float32).copy()
        nan_cam[3] = np.nan                          # l1.x: a fan of rays with one NaN direction component
        far_cam = np.array(CAMERAS["K0"], dtype=np.float32).copy()
        far_cam[0:3] = (3.5, 1.25, 2.75)             # outside the cube: misses, grazing entries
        ctx.set_pipeline(1)
        other.set_pipeline(1)
        for cam in (CAMERAS["K1"], CAMERAS["K2"], nan_cam, far_cam):
            a = ctx.render(pool, 1920, 1080, cam, 7, mode, bounces=3)
            b = other.render(pool, 1920, 1080, cam, 7, mode, bounces=3)
            assert np.array_equal(a["rgba"], b["rgba"])
            assert np.array_equal(a["depth"].view(np.uint32), b["depth"].view(np.uint32))
            assert a["hits"].tobytes() == b["hits"].tobytes()
    finally:
        other.close()


@pytest.mark.gpu
def test_asm_loop_equals_cxx_loop_under_varied_occupancy(ctx):
    """Same cross-check, repeated with different numbers of persistent waves per CU, refill thresholds and frame
    numbers: a missed wait state or an unwaited load in the assembly would show up as a rare, load-dependent mismatch."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    other = hiplib.HipContext(0, lib_path=hiplib.CXXLOOP_LIB_PATH)
    try:
        pool, _ = scene.build_scene(2048)
        ctx.set_pipeline(1)
        other.set_pipeline(1)
        ctx.pool_upload(pool)
        other.pool_upload(pool)
        ref = {}
        for k, (waves, thresh) in enumerate([(1, 6), (2, 12), (4, 8), (7, 10), (10, 9), (13, 4), (16, 15), (20, 8), (0, 0)] * 2):
            frame = 2 + (k % 5)
            mode = (0, 2)[k % 2]
            ctx.set_tuning(waves, thresh)
            a = ctx.render(None, 1280, 720, CAMERAS["K1"], frame, mode, bounces=3)
            key = (frame, mode)
            if key not in ref:
                ref[key] = other.render(None, 1280, 720, CAMERAS["K1"], frame, mode, bounces=3)
            b = ref[key]
            assert np.array_equal(a["rgba"], b["rgba"]), (waves, thresh, frame, mode)
            assert np.array_equal(a["depth"].view(np.uint32), b["depth"].view(np.uint32)), (waves, thresh, frame, mode)
            assert a["hits"].tobytes() == b["hits"].tobytes(), (waves, thresh, frame, mode)
    finally:
        ctx.set_tuning(0, 0)
        other.close()
