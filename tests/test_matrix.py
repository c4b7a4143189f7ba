"""The cells of the round-6 measurement matrix (profiles/round6_matrix.md) against the reference shader's own output
(llvmpipe, tests/golden/make_golden_matrix.py): SURVEY 8(d)'s cameras K0 / K1 / K2 over three scenes -- the default
terrain (K0 / K1: tests/test_config3.py), a second terrain (seed 2 at amplitude 18, the largest at which no camera is
under ground) and the new family with real 3-D structure ("caves": scene/svo_scene.c family 1) -- at the benchmark's full
size (8192^3, 1920x1080, renderMode 0, every 8th pixel in x and y), plus the new family at 128^3 / 256^3 in every render
mode, whole images.  CPU legs: the oracle.  GPU legs: the HIP pipelines through the C ABI, the persistent pipeline also with
the library's default ring (6 submissions in flight x 4 frames)."""
import os
import zlib

import numpy as np
import pytest

import poolcache

import helpers
import svo_raytracer_amd.scene as scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "matrix_golden.npz")
_z = {}


def Z():
    if "z" not in _z:
        _z["z"] = np.load(GOLD)
    return _z["z"]


def _small():
    return [tuple(str(s).split(":")) for s in Z()["index_small"]]


def _full(scene_key):
    return [n for n, sk in (str(s).split(":") for s in Z()["index_full"]) if sk == scene_key]


SCENE_KEYS = ["t1a8", "t2a18", "c1a8d64", "d1a8x24"]


@pytest.fixture(scope="module", params=SCENE_KEYS)
def big(request):
    """(scene key, the 8192^3 pool of that scene), one at a time (a pool is 1.4 - 2.0 GB)"""
    z = Z()
    sk = request.param
    family, seed, amp, dens = (str(v) for v in z[sk + "/scene"])
    pool = poolcache.pool(family, 8192, int(seed), int(amp), int(dens))
    assert pool.size == int(z[sk + "/pool_size"][0]) and zlib.crc32(pool.tobytes()) == int(z[sk + "/pool_crc32"][0]), \
        "scene generator drifted: regenerate tests/golden/matrix_golden.npz"
    yield sk, pool


def _small_case(name, pk):
    z = Z()
    w, h, frame, mode, same = [int(v) for v in z[name + "/meta"]]
    return {"pool": z["pool/" + pk], "w": w, "h": h, "frame": frame, "mode": mode, "cam": z[name + "/cam"],
            "rgba": z[name + "/rgba"], "depth_bits": z[name + "/depth_bits"], "first_hit": z[name + "/first_hit"],
            "patched_same": bool(same)}


def _check_small(res, g, name):
    m = helpers.compare_with_golden(res, g)
    assert all(v == 0 for v in m.values()), (name, m)


def _check_full(res, z, name, step):
    sub = (slice(0, res["rgba"].shape[0], step), slice(0, res["rgba"].shape[1], step))
    fh = z[name + "/first_hit"]
    hit = fh[..., 0] != 0
    assert (res["rgba"][sub] == z[name + "/rgba"]).all(), name
    assert (res["depth"].view(np.uint32)[sub] == z[name + "/depth_bits"]).all(), name
    h = res["hits"]
    assert (h["pointer"][sub] == fh[..., 0]).all(), name
    assert ((h["value"][sub] == fh[..., 1]) | ~hit).all(), name
    assert ((h["raw_normal"][sub] == fh[..., 2]) | ~hit).all(), name
    assert ((h["level"][sub] == (fh[..., 3] >> 16)) | ~hit).all(), name
    assert ((h["iter"][sub] == (fh[..., 3] & 0xFFFF)) | ~hit).all(), name


# ------------------------------------------------------------------------------------------------ CPU: the oracle

def test_fixture_holds_what_the_matrix_needs():
    z = Z()
    assert {n.split("_")[1] for n in _full("t2a18")} == {"K0", "K1", "K2"}
    assert {n.split("_")[1] for n in _full("c1a8d64")} >= {"K0", "K1", "K2"}
    assert _full("t1a8") == ["t1a8_K2_f2"]      # K0 / K1 over the default terrain: config3_8192.npz
    for name, pk in _small():
        assert int(z[name + "/meta"][4]) == 1, name      # the instrumented shader == the unmodified one
    # every cell sees the scene: thousands of primary hits in the subsample
    for sk in SCENE_KEYS:
        for name in _full(sk):
            assert int((z[name + "/first_hit"][..., 0] != 0).sum()) > 5000, name


@pytest.mark.parametrize("name,pk", _small())
def test_oracle_matches_the_reference_shader_on_the_caves_family(name, pk):
    from oracle import oracle
    g = _small_case(name, pk)
    res = oracle.render(g["pool"], g["w"], g["h"], g["cam"], g["frame"], g["mode"])
    _check_small(res, g, name)


def test_oracle_matches_the_reference_shader_on_the_matrix_cells(big):
    from oracle import oracle
    sk, pool = big
    z = Z()
    step = int(z["step"][0])
    for name in _full(sk):
        w, h, frame, mode, bounces, mirror = (int(v) for v in z[name + "/meta"])
        res = oracle.render(pool, w, h, z[name + "/cam"], frame, mode, bounces=bounces, xstep=step, ystep=step)
        _check_full(res, z, name, step)


# ------------------------------------------------------------------------------------------------ GPU: the HIP path

@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_hip_matches_the_reference_shader_on_the_caves_family(pipeline):
    from svo_raytracer_amd import hiplib
    ctx = helpers.DualContext()
    try:
        ctx.set_pipeline(pipeline)
        last = None
        for name, pk in _small():
            g = _small_case(name, pk)
            res = ctx.render(g["pool"] if pk != last else None, g["w"], g["h"], g["cam"], g["frame"], g["mode"])
            last = pk
            _check_small(res, g, name)
        if pipeline == 1:
            assert ctx.derived_info()["walkable"]      # the family's pools take the descriptor walk
    finally:
        ctx.close()


@pytest.mark.gpu
def test_hip_matches_the_reference_shader_on_the_matrix_cells(big):
    """every cell of the scene: one dispatch at a time on all three pipelines, then -- pipeline 1, no tuning call -- inside the
    library's default ring: 6 submissions in flight x 4 frames, the golden frame somewhere in the middle of them"""
    from svo_raytracer_amd import hiplib
    sk, pool = big
    z = Z()
    step = int(z["step"][0])
    ctx = helpers.DualContext()
    try:
        ctx.pool_upload(pool)
        for name in _full(sk):
            w, h, frame, mode, bounces, mirror = (int(v) for v in z[name + "/meta"])
            for pipeline in (1, 0, 2):
                ctx.set_pipeline(pipeline)
                _check_full(ctx.render(None, w, h, z[name + "/cam"], frame, mode, bounces=bounces), z, name, step)
            if mode != 0:
                continue
            ctx.set_pipeline(1)
            assert ctx.derived_info()["walkable"]
            ctx.set_hit_records(True)
            ctx.set_params(2, mode, 0, 0, bounces, 0, 1)
            ctx.ring_create(6, 4, want_hits=True)
            first = max(1, frame - 9)      # the golden frame lands in the 1st .. 3rd of the six submissions
            slots = [ctx.ring_submit(first + 4 * b, 4) for b in range(6)]
            b, k = divmod(frame - first, 4)
            assert ctx.launch_info()["waves_per_cu"] == 10
            ctx.ring_wait(slots[b])
            _check_full(ctx.ring_read(slots[b], k, want_hits=True), z, name, step)
            for s in slots:
                ctx.ring_wait(s)
            ctx.ring_destroy()
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------ the beam pre-pass on the new family
# (floating debris and overhangs are what a conservative coarse depth pass must survive: a 4x4 block's pyramid may graze a ball
# that none of its sixteen rays' neighbours on the terrain sees)

def _beam_bad(res, g):
    bad = helpers.compare_with_golden(res, g)
    bad.pop("iter")                       # fewer iterations is the point
    if g["mode"] == 1:
        bad.pop("rgba")                   # renderMode 1 displays the iteration count
    return bad


@pytest.mark.parametrize("name,pk", _small())
def test_oracle_with_beam_matches_the_reference_shader_on_the_caves_family(name, pk):
    from oracle import oracle
    g = _small_case(name, pk)
    res = oracle.render(g["pool"], g["w"], g["h"], g["cam"], g["frame"], g["mode"], use_beam=True)
    bad = _beam_bad(res, g)
    assert bad == {k: 0 for k in bad}, (name, bad)


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1])
def test_hip_with_beam_matches_the_reference_shader_on_the_caves_family(pipeline):
    from oracle import oracle
    ctx = helpers.DualContext()
    try:
        ctx.set_pipeline(pipeline)
        last = None
        for name, pk in _small():
            g = _small_case(name, pk)
            res = ctx.render(g["pool"] if pk != last else None, g["w"], g["h"], g["cam"], g["frame"], g["mode"], use_beam=1)
            last = pk
            bad = _beam_bad(res, g)
            assert bad == {k: 0 for k in bad}, (name, bad)
            ref = oracle.render(g["pool"], g["w"], g["h"], g["cam"], g["frame"], g["mode"], use_beam=True)
            assert np.array_equal(ctx.read_beam().view(np.uint32), ref["beam"].view(np.uint32)), name     # the coarse pass, bit for bit
            assert (res["hits"]["iter"] == ref["hits"]["iter"]).all(), name
    finally:
        ctx.close()


@pytest.mark.gpu
def test_hip_with_beam_on_the_full_size_caves_scene():
    """8192^3 caves, 1080p, K1 and K2, mode 0: every byte of the frame without the beam pass except the iteration counts"""
    from svo_raytracer_amd.cameras import CAMERAS
    ctx = helpers.DualContext()
    try:
        ctx.set_pipeline(1)
        ctx.pool_upload(poolcache.pool("caves", 8192, 1, 8, 64))
        for cam in ("K1", "K2"):
            plain = ctx.render(None, 1920, 1080, CAMERAS[cam], 2, 0)
            beam = ctx.render(None, 1920, 1080, CAMERAS[cam], 2, 0, use_beam=1)
            assert (beam["rgba"] == plain["rgba"]).all() and (beam["depth"].view(np.uint32) == plain["depth"].view(np.uint32)).all()
            for k in ("pointer", "value", "raw_normal", "level"):
                assert (beam["hits"][k] == plain["hits"][k]).all(), (cam, k)
            assert (beam["hits"]["t"].view(np.uint32) == plain["hits"]["t"].view(np.uint32)).all()
            assert int(beam["hits"]["iter"].sum()) < int(plain["hits"]["iter"].sum())
    finally:
        ctx.close()
