"""BASELINE config 5 in the reference's own terms: "64 spp accumulated path-traced GI" = 64 frames of the cross-frame
accumulation of svotrace.comp:712-719 (frameNumber 2..65, Main.java:16,275) on one persistent rgba8 framebuffer, 8192^3,
1920x1080, renderMode 0.  tests/golden/c5_progressive.npz holds every 8th pixel of frames 2, 3, 33 and 65 of that sequence
as the reference shader itself renders it under llvmpipe (make_golden_c5.py: the commented block switched on in memory).

CPU: the oracle's statement of the accumulation reproduces it.  GPU: svo_set_progressive + svo_set_sequence -- the whole
sequence in ONE persistent launch, the recurrence applied in frame order afterwards -- leaves the same bytes, and so do
64 dispatches of one frame each, a sequence continued on the image a shorter one left, and sequences in flight on a ring."""
import os

import numpy as np
import pytest
import helpers

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    z = np.load(os.path.join(HERE, "golden", "c5_progressive.npz"))
    n, w, h, mode = (int(v) for v in z["meta"])
    return dict(z=z, n=n, w=w, h=h, mode=mode, step=int(z["step"][0]), cam=z["cam"], keep=[int(v) for v in z["keep"]],
                frames=[int(v) for v in z["frames"]])


@pytest.fixture(scope="module")
def pool(golden):
    import zlib
    import svo_raytracer_amd.scene as scene
    p, _ = scene.build_scene(golden["n"])
    assert p.size == int(golden["z"]["pool_size"][0]) and zlib.crc32(p.tobytes()) == int(golden["z"]["pool_crc32"][0])
    return p


def _sub(img, step):
    return img[::step, ::step]


def test_fixture_is_the_sequence_it_claims(golden):
    assert golden["frames"] == list(range(2, 66)) and golden["keep"] == [2, 3, 33, 65]
    assert (golden["w"], golden["h"], golden["mode"], golden["n"]) == (1920, 1080, 0, 8192)
    z = golden["z"]
    # the image converges: frame 2 on a zeroed image is a third of a sample, frame 65 nearly the mean
    m = [float(z["f%d/rgba" % f][..., :3].mean()) for f in golden["keep"]]
    assert m[0] < m[1] < m[2] and abs(m[3] - m[2]) < 3.0 and m[0] < 0.4 * m[3]


def test_oracle_reproduces_the_reference_s_64_frame_accumulation(golden, pool):
    from oracle import oracle
    g, z, st = golden, golden["z"], golden["step"]
    last = np.zeros((g["h"], g["w"], 4), dtype=np.uint8)      # "fresh": the image glTexStorage2D left
    for f in g["frames"]:
        r = oracle.render(pool, g["w"], g["h"], g["cam"], f, g["mode"], xstep=st, ystep=st, want_hits=False, last_rgba=last)
        last = r["rgba"]
        if f in g["keep"]:
            assert np.array_equal(_sub(last, st), z["f%d/rgba" % f]), f
            assert np.array_equal(_sub(r["depth"].view(np.uint32), st), z["f%d/depth_bits" % f]), f


@pytest.fixture(scope="module")
def ctx(pool):
    from svo_raytracer_amd import hiplib
    c = helpers.DualContext()
    c.pool_upload(pool)
    yield c
    c.close()


def _check(got, z, f, st):
    assert np.array_equal(_sub(got["rgba"], st), z["f%d/rgba" % f]), ("rgba", f)
    assert np.array_equal(_sub(got["depth"].view(np.uint32), st), z["f%d/depth_bits" % f]), ("depth", f)


@pytest.mark.gpu
def test_sequence_in_one_launch_equals_the_reference_shader(golden, ctx):
    g, z, st = golden, golden["z"], golden["step"]
    ctx.set_pipeline(1)
    ctx.resize(g["w"], g["h"])
    ctx.set_camera(g["cam"])
    ctx.set_progressive(True)
    try:
        whole = {}
        for f in g["keep"]:                       # frames 2..f as ONE dispatch on a zeroed image
            ctx.set_sequence(f - 1, fresh=True)
            ctx.set_params(2, g["mode"], 0, 0, 2, 0, 1)
            ctx.dispatch()
            whole[f] = {"rgba": ctx.read_color(), "depth": ctx.read_depth()}
            _check(whole[f], z, f, st)
        # continued: frames 2..33 fresh, then frames 34..65 on the image they left
        ctx.set_sequence(32, fresh=True)
        ctx.set_params(2, g["mode"], 0, 0, 2, 0, 1)
        ctx.dispatch()
        ctx.set_sequence(32, fresh=False)
        ctx.set_params(34, g["mode"], 0, 0, 2, 0, 1)
        ctx.dispatch()
        got = {"rgba": ctx.read_color(), "depth": ctx.read_depth()}
        assert np.array_equal(got["rgba"], whole[65]["rgba"]) and np.array_equal(got["depth"].view(np.uint32), whole[65]["depth"].view(np.uint32))
        # one dispatch per frame, the reference's own loop: every pixel of the final image and of frame 33
        ctx.set_sequence(1, fresh=True)
        ctx.set_params(2, g["mode"], 0, 0, 2, 0, 1)
        ctx.dispatch()                            # (fresh applies to a sequence of one as well)
        ctx.set_sequence(1, fresh=False)
        for f in range(3, 66):
            ctx.set_params(f, g["mode"], 0, 0, 2, 0, 1)
            ctx.dispatch()
            if f in (33, 65):
                one = {"rgba": ctx.read_color(), "depth": ctx.read_depth()}
                assert np.array_equal(one["rgba"], whole[f]["rgba"]), f
                assert np.array_equal(one["depth"].view(np.uint32), whole[f]["depth"].view(np.uint32)), f
    finally:
        ctx.set_sequence(1, fresh=False)
        ctx.set_progressive(False)


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 2])
def test_sequences_on_the_other_pipelines_fall_back_to_one_launch_per_frame(golden, ctx, pipeline):
    g, z, st = golden, golden["z"], golden["step"]
    ctx.set_pipeline(pipeline)
    ctx.resize(g["w"], g["h"])
    ctx.set_camera(g["cam"])
    ctx.set_progressive(True)
    try:
        ctx.set_sequence(2, fresh=True)
        ctx.set_params(2, g["mode"], 0, 0, 2, 0, 1)
        ctx.dispatch()
        _check({"rgba": ctx.read_color(), "depth": ctx.read_depth()}, z, 3, st)
    finally:
        ctx.set_sequence(1, fresh=False)
        ctx.set_progressive(False)
        ctx.set_pipeline(1)


@pytest.mark.gpu
def test_sequences_in_flight_on_the_ring_through_jni_typed_calls(golden, ctx):
    """what bench.py --config C5 runs: several 64-frame sequences in flight, each in a ring slot of its own"""
    import ctypes
    from svo_raytracer_amd import hiplib
    g, z, st = golden, golden["z"], golden["step"]
    L = hiplib.lib()
    h = ctx._h
    J = lambda name: getattr(L, "Java_src_engine_HipRenderer_" + name)   # noqa: E731
    i32, i64 = ctypes.c_int32, ctypes.c_int64
    for name, args in (("nSetSequence", [i64, i32, i32]), ("nRingSubmit", [i64, i32, i32]), ("nRingCreate", [i64, i32, i32, i32]),
                       ("nSetProgressive", [i64, i32]), ("nRingWait", [i64, i32]), ("nRingDestroy", [i64])):
        J(name).argtypes = [ctypes.c_void_p, ctypes.c_void_p] + args
        J(name).restype = i32
    ctx.set_pipeline(1)
    ctx.set_tuning(10, 9)
    ctx.resize(g["w"], g["h"])
    ctx.set_camera(g["cam"])
    ctx.set_params(2, g["mode"], 0, 0, 2, 0, 1)
    try:
        assert J("nSetProgressive")(None, None, h.value, 1) == 0
        assert J("nSetSequence")(None, None, h.value, 64, 1) == 0
        assert J("nRingCreate")(None, None, h.value, 3, 1, 0) == 0
        slots = [J("nRingSubmit")(None, None, h.value, 2, 1) for _ in range(3)]
        assert slots == [0, 1, 2]
        for s in slots:
            assert J("nRingWait")(None, None, h.value, s) == 0
            _check(ctx.ring_read(s, 0), z, 65, st)
        assert J("nRingDestroy")(None, None, h.value) == 0
    finally:
        ctx.set_sequence(1, fresh=False)
        ctx.set_progressive(False)
        ctx.set_tuning(0, 0)
