"""the beam pre-pass (useBeamOptimization, Main.java:51, 257-266; svobeam.comp:617-637;
svotrace.comp:438, 656-658).  The reference's own coarse pass is dormant and inconsistent, so the parity statement
is the one a working version has to satisfy: with use_beam = 1 every byte of the colour and depth images and the hit
records' pointer / value / normal / level / t equal those with use_beam = 0 -- which the reference shader itself
pinned (llvmpipe goldens) -- while iteration counts drop.  The CPU statement of the pass (oracle/svo_oracle.c::
svo_oracle_beam) is checked against the goldens here on the CPU; the HIP pass is checked against both on the GPU."""
import os

import numpy as np
import pytest

import poolcache

from helpers import compare_with_golden, golden_case, golden_cases
import helpers

PIPELINES = [int(v) for v in os.environ.get("SVO_TEST_PIPELINES", "0,1,2").split(",")]


def _check_against_golden(res, g):
    bad = compare_with_golden(res, g)
    bad.pop("iter")                       # fewer iterations is the point
    if g["mode"] == 1:
        bad.pop("rgba")                   # renderMode 1 displays the iteration count
    return bad


@pytest.mark.parametrize("name,poolkey", golden_cases())
def test_oracle_with_beam_matches_the_reference_shader_goldens(name, poolkey):
    from oracle import oracle
    g = golden_case(name, poolkey)
    res = oracle.render(g["pool"], g["w"], g["h"], g["cam"], g["frame"], g["mode"], use_beam=True)
    bad = _check_against_golden(res, g)
    assert bad == {k: 0 for k in bad}, bad
    plain = oracle.render(g["pool"], g["w"], g["h"], g["cam"], g["frame"], g["mode"])
    assert (res["hits"]["t"].view(np.uint32) == plain["hits"]["t"].view(np.uint32)).all()
    assert res["stats"]["iterations"] <= plain["stats"]["iterations"]


def test_beam_distance_is_conservative_and_useful():
    """Every primary hit lies behind its block's start distance; on open terrain the pass removes a third or more of
    the primary iterations."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    pool, _ = scene.build_scene(512)
    for cam in ("K0", "K1", "K2"):
        plain = oracle.render(pool, 256, 144, CAMERAS[cam], 2, 3)
        beam = oracle.beam(pool, 256, 144, CAMERAS[cam])
        per_px = np.repeat(np.repeat(beam, 4, axis=0), 4, axis=1)[:144, :256]
        hit = plain["hits"]["pointer"] != 0
        assert (plain["hits"]["t"][hit] > per_px[hit]).all()
        fast = oracle.render(pool, 256, 144, CAMERAS[cam], 2, 3, use_beam=True)
        assert fast["stats"]["iterations"] < 0.7 * plain["stats"]["iterations"], cam
        assert (fast["hits"]["pointer"] == plain["hits"]["pointer"]).all()


def test_beam_falls_back_on_cameras_that_are_not_a_planar_rectangle():
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    pool, _ = scene.build_scene(64)
    cam = np.array(CAMERAS["K1"], dtype=np.float32).copy()
    cam[12] += 0.3                         # r2 off the plane of the other three corners
    assert (oracle.beam(pool, 64, 48, cam) == 0).all()
    cam = np.array(CAMERAS["K1"], dtype=np.float32).copy()
    cam[4] = np.nan
    assert (oracle.beam(pool, 64, 48, cam) == 0).all()


# ---- GPU -----------------------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def ctx():
    from svo_raytracer_amd import hiplib
    c = helpers.DualContext()
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", PIPELINES)
@pytest.mark.parametrize("name,poolkey", golden_cases())
def test_hip_with_beam_matches_goldens_and_the_oracle_pass(ctx, name, poolkey, pipeline):
    from oracle import oracle
    g = golden_case(name, poolkey)
    ctx.set_pipeline(pipeline)
    res = ctx.render(g["pool"], g["w"], g["h"], g["cam"], g["frame"], g["mode"], use_beam=1)
    bad = _check_against_golden(res, g)
    assert bad == {k: 0 for k in bad}, bad
    ref = oracle.render(g["pool"], g["w"], g["h"], g["cam"], g["frame"], g["mode"], use_beam=True)
    assert np.array_equal(ctx.read_beam().view(np.uint32), ref["beam"].view(np.uint32))     # the coarse pass, bit for bit
    assert (res["hits"]["iter"] == ref["hits"]["iter"]).all()                                 # and the shortened walks
    assert (res["rgba"] == ref["rgba"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", PIPELINES)
def test_config3_with_beam_equals_config3_without(ctx, pipeline):
    """BASELINE config 3 (8192^3, 1920x1080, primary + bounce): every output byte except the iteration counts."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    pool = poolcache.pool()
    ctx.set_pipeline(pipeline)
    for cam in ("K1", "K2"):
        plain = ctx.render(pool if cam == "K1" else None, 1920, 1080, CAMERAS[cam], 2, 0)
        fast = ctx.render(None, None, None, None, 2, 0, use_beam=1)
        assert np.array_equal(plain["rgba"], fast["rgba"])
        assert np.array_equal(plain["depth"].view(np.uint32), fast["depth"].view(np.uint32))
        for k in ("pointer", "value", "raw_normal", "level"):
            assert np.array_equal(plain["hits"][k], fast["hits"][k]), k
        assert np.array_equal(plain["hits"]["t"].view(np.uint32), fast["hits"]["t"].view(np.uint32))
        it0, it1 = int(plain["hits"]["iter"].sum()), int(fast["hits"]["iter"].sum())
        assert it1 < it0
        step = 24
        ref = oracle.render(pool, 1920, 1080, CAMERAS[cam], 2, 0, xstep=step, ystep=step, use_beam=True)
        sub = (slice(0, 1080, step), slice(0, 1920, step))
        assert (ref["hits"]["iter"][sub] == fast["hits"]["iter"][sub]).all()
        assert np.array_equal(ctx.read_beam().view(np.uint32), ref["beam"].view(np.uint32))
        print("pipeline %d camera %s: primary iterations %d -> %d (%.1f %%)" % (pipeline, cam, it0, it1, 100.0 * it1 / it0))


@pytest.mark.gpu
def test_beam_with_stripes_samples_and_odd_sizes(ctx):
    """Packed stripes (multi-GPU sharding) only fill the block rows under their tile rows; an image whose size is not
    a multiple of 4; several samples per pixel; a camera the pass refuses (falls back to 0)."""
    import torch
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from svo_raytracer_amd.tiles import stripe_layout, deinterleave
    pool, _ = scene.build_scene(256)
    w, h, world = 202, 117, 3
    for pipeline in PIPELINES:
        ctx.set_pipeline(pipeline)
        full = ctx.render(pool, w, h, CAMERAS["K1"], 2, 0, spp=2)
        fast = ctx.render(None, None, None, None, 2, 0, spp=2, use_beam=1)
        assert np.array_equal(full["rgba"], fast["rgba"]) and np.array_equal(full["hits"]["pointer"], fast["hits"]["pointer"])
        rpr = stripe_layout(h, world, 0)[4]
        col = torch.zeros((rpr * world, w), dtype=torch.int32, device="cuda")
        dep = torch.zeros((rpr * world, w), dtype=torch.float32, device="cuda")
        ctx.bind_outputs(col.data_ptr(), dep.data_ptr(), None)
        ctx.set_params(2, 0, 0, 1, 2, 0, 2)
        for r in range(world):
            first, step, n, out0, rows = stripe_layout(h, world, r)
            ctx.set_stripes(first, step, n, out0)
            ctx.dispatch()
        torch.cuda.synchronize()
        ctx.bind_outputs(None, None, None)
        ctx.set_rows(0, h)
        got = deinterleave(col.cpu().numpy().view(np.uint8).reshape(rpr * world, w, 4), world, rpr, h)
        assert (got == full["rgba"]).all()
        bad_cam = np.array(CAMERAS["K1"], dtype=np.float32).copy()
        bad_cam[12] += 0.3
        a = ctx.render(None, None, None, bad_cam, 2, 2)
        b = ctx.render(None, None, None, bad_cam, 2, 2, use_beam=1)
        assert (ctx.read_beam() == 0).all()
        assert np.array_equal(a["rgba"], b["rgba"]) and a["hits"].tobytes() == b["hits"].tobytes()
