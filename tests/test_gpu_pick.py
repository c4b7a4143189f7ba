"""The crosshair pick without waiting for its frame, and svo_dispatch_async's two alternating {stream, image} sets
(include/svo_hip.h, svo_set_pick / svo_set_overlap; the reference: Main.updateEarly reads ONE pixel of the previous frame's depth
image before it dispatches the next frame, Main.java:132-146, 257-289).  The values must be the image's; the images must be the
bytes one stream and one image set give; the pick must really come from the mail where the header says it does."""
import ctypes

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from svo_raytracer_amd import hiplib
    c = helpers.DualContext()
    yield c
    c.close()


def _frame(ctx):
    return {"rgba": ctx.read_color(), "depth": ctx.read_depth(), "hits": ctx.read_hits()}


def _pixel_matches(px, img, x, y, hits=True):
    rgba, depth, hit = px
    assert (rgba == img["rgba"][y, x]).all()
    assert np.float32(depth).view(np.uint32) == img["depth"].view(np.uint32)[y, x]
    if hits:
        for k in ("pointer", "value", "raw_normal", "level", "iter"):
            assert hit[k] == img["hits"][k][y, x], k
        assert hit["t"].view(np.uint32) == img["hits"]["t"].view(np.uint32)[y, x]


@pytest.mark.parametrize("case", [c for c in helpers.golden_cases() if c[0] in (
    "s128_K1_m0", "s128_K2_m2", "s128_K0_m3", "s64_K0_m1", "s128k6_K1_m0", "dust256_KDUST_m2", "s64_K0_m2_odd", "s64sdf_KEDIT_m0")])
def test_pick_equals_the_image_on_the_goldens(ctx, case):
    """the crosshair (image centre, the default pick) of a reference-shader golden: answered from the mail, equal to the image
    the same dispatch leaves, which equals the shader's"""
    g = helpers.golden_case(*case)
    ctx.pool_upload(g["pool"])
    ctx.resize(g["w"], g["h"])
    ctx.set_camera(g["cam"])
    ctx.set_params(g["frame"], g["mode"], 0, 0, 2, 0, 1)
    info0 = ctx.pick_info()
    assert (info0["x"], info0["y"]) == (g["w"] // 2, g["h"] // 2)
    ctx.dispatch_async()
    px = ctx.read_pixel(g["w"] // 2, g["h"] // 2)
    info1 = ctx.pick_info()
    assert info1["from_mail"] == info0["from_mail"] + 1 and info1["waited"] == info0["waited"]
    img = _frame(ctx)
    _pixel_matches(px, img, g["w"] // 2, g["h"] // 2)
    bad = helpers.compare_with_golden(img, g)
    assert all(v == 0 for v in bad.values()), bad
    # any other position: the waiting path, the same image
    px2 = ctx.read_pixel(3, 5)
    assert ctx.pick_info()["waited"] == info1["waited"] + 1
    _pixel_matches(px2, img, 3, 5)


def test_pick_positions_and_the_frames_that_carry_none(ctx):
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene3(256, 2, 8, 128)
    w, h = 200, 120
    ctx.pool_upload(pool)
    ctx.resize(w, h)
    ctx.set_camera(CAMERAS["K1"])
    ctx.set_params(2, 0, 0, 0, 2, 0, 1)
    ctx.dispatch()
    want = _frame(ctx)
    n_mail = ctx.pick_info()["from_mail"]
    # every corner of the image, the debug square (svotrace.comp:696-700), tiles in every screen band, both column directions
    for i, (x, y) in enumerate([(0, 0), (3, 3), (w - 1, h - 1), (0, h - 1), (w - 1, 0), (100, 60), (17, 93), (150, 8), (64, 64), (199, 50)]):
        ctx.set_pick(x, y)
        for rep in range(2):       # consecutive launches walk the columns in opposite directions
            ctx.dispatch_async()
            px = ctx.read_pixel(x, y)
            n_mail += 1
            assert ctx.pick_info()["from_mail"] == n_mail, (x, y)
            _pixel_matches(px, want, x, y)
            got = _frame(ctx)
            assert (got["rgba"] == want["rgba"]).all() and (got["depth"].view(np.uint32) == want["depth"].view(np.uint32)).all()
            assert (got["hits"] == want["hits"]).all()
    with pytest.raises(Exception):
        ctx.set_pick(w, 0)
    # frames that get no pick launch answer through the waiting path -- same values; the other pipelines' frames get one
    # (pipeline 2 lives in the variants library: a context -- and counters -- of its own behind DualContext)
    ctx.set_pick(100, 60)

    def after(setup, undo, mail, wait, hits=True):
        setup()
        i0 = ctx.pick_info()
        ctx.dispatch_async()
        px = ctx.read_pixel(100, 60)
        i1 = ctx.pick_info()
        assert (i1["from_mail"] - i0["from_mail"], i1["waited"] - i0["waited"]) == (mail, wait)
        undo()
        _pixel_matches(px, want, 100, 60, hits)

    for p in (0, 2):
        after(lambda: ctx.set_pipeline(p), lambda: ctx.set_pipeline(1), 1, 0)
    after(lambda: ctx.set_pick(-1, -1), lambda: ctx.set_pick(100, 60), 0, 1)
    after(lambda: ctx.set_rows(56, 72), lambda: ctx.resize(w + 8, h) or ctx.resize(w, h), 0, 1)      # a row band: no pick launch
    after(lambda: ctx.set_params(2, 0, 0, 1, 2, 0, 1), lambda: ctx.set_params(2, 0, 0, 0, 2, 0, 1), 0, 1, hits=False)   # the beam pre-pass
                                                                                   # (its frames differ in iteration counts only)
    n_mail = ctx.pick_info()["from_mail"]
    ctx.set_pick(100, 60)
    # without hit records the pick still answers colour and depth from the mail
    ctx.set_hit_records(False)
    ctx.dispatch_async()
    rgba, depth = np.zeros(4, np.uint8), np.zeros(1, np.float32)
    assert ctx._L.svo_read_pixel(ctx._h, 100, 60, rgba.ctypes.data, depth.ctypes.data, None) == 0
    assert ctx.pick_info()["from_mail"] == n_mail + 1
    assert (rgba == want["rgba"][60, 100]).all() and depth.view(np.uint32)[0] == want["depth"].view(np.uint32)[60, 100]
    ctx.set_hit_records(True)


def test_overlapped_dispatches_leave_the_bytes_of_one_stream(ctx):
    """frames 2..13 dispatched back to back (four image sets in turn, up to four frames in flight) against the same frames on one
    stream and one image set; a read-back always names the LAST dispatched frame"""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(512)
    w, h = 640, 360
    ctx.pool_upload(pool)
    ctx.resize(w, h)
    ctx.set_camera(CAMERAS["K1"])
    ctx.set_overlap(False)
    single = []
    for f in range(2, 14):
        ctx.set_params(f, 0, 0, 0, 2, 0, 1)
        ctx.dispatch_async()
        single.append(_frame(ctx))
    p0 = ctx.output_device_ptrs()
    ctx.set_overlap(True)
    ptrs = set()
    for i, f in enumerate(range(2, 14)):
        ctx.set_params(f, 0, 0, 0, 2, 0, 1)
        ctx.dispatch_async()
        ptrs.add(ctx.output_device_ptrs()[0])
        if i % 3 == 2:          # read some frames while the next ones are not yet dispatched, skip others entirely
            got = _frame(ctx)
            for k in ("rgba", "hits"):
                assert (got[k] == single[i][k]).all(), (f, k)
            assert (got["depth"].view(np.uint32) == single[i]["depth"].view(np.uint32)).all(), f
    assert len(ptrs) == 4 and p0[0] in ptrs          # four image sets took turns (the default)
    ctx.sync()
    got = _frame(ctx)
    assert (got["rgba"] == single[-1]["rgba"]).all() and (got["hits"] == single[-1]["hits"]).all()
    # the waiting dispatch, the counting pass and the accumulation go on from the current set
    ctx.set_params(5, 0, 0, 0, 2, 0, 1)
    ctx.dispatch()
    got = _frame(ctx)
    assert (got["rgba"] == single[3]["rgba"]).all()
    st = ctx.count_frame()
    assert st["rays"] > w * h
    ctx.set_progressive(True)
    imgs = []
    for f in (2, 3, 4):
        ctx.set_params(f, 0, 0, 0, 2, 0, 1)
        ctx.dispatch_async()
        imgs.append(ctx.read_color())
    ctx.set_progressive(False)
    from oracle import oracle
    last = single[3]["rgba"]       # the image the accumulation starts on: the current set's (frame 5 above, then the counting pass = frame 5)
    for f, im in zip((2, 3, 4), imgs):
        ref = oracle.render(pool, w, h, CAMERAS["K1"], f, 0, want_hits=False, last_rgba=last, rows=(100, 108))
        assert (im[100:108] == ref["rgba"][100:108]).all(), f
        last = im


def test_the_reference_loop_through_jni_typed_calls(ctx):
    """Main.updateEarly as a drop-in host runs it: per frame nSetCamera, nSetParams, nDispatchAsync, then the crosshair of THAT
    frame (nReadPixel at the image centre) before the next dispatch -- every pick from the mail, every value the oracle's"""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import orbit_path
    from oracle import oracle
    L = hiplib.lib()
    vp, jint, jlong, jfloat = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float

    def fn(name, res, *a):
        f = getattr(L, "Java_src_engine_HipRenderer_" + name)
        f.restype = res
        f.argtypes = [vp, vp] + list(a)
        return lambda *v: f(None, None, *v)

    nCreate, nDestroy = fn("nCreate", jlong, jint), fn("nDestroy", jint, jlong)
    nPoolUpload = fn("nPoolUpload", jint, jlong, jlong, jlong)
    nSetCamera = fn("nSetCamera", jint, jlong, *([jfloat] * 15))
    nSetParams = fn("nSetParams", jint, jlong, *([jint] * 7))
    nResize = fn("nResize", jint, jlong, jint, jint)
    nDispatchAsync, nSync = fn("nDispatchAsync", jint, jlong), fn("nSync", jint, jlong)
    nReadPixel = fn("nReadPixel", jint, jlong, jint, jint, jlong, jlong, jlong)
    nReadDepth = fn("nReadDepth", jint, jlong, jlong)
    nPickInfo = fn("nPickInfo", jlong, jlong, jlong, jlong)
    nSetPick = fn("nSetPick", jint, jlong, jint, jint)
    pool, _ = scene.build_scene(512)
    w, h, n = 640, 360, 40
    cams, fns = orbit_path(n)
    j = nCreate(0)
    assert j
    try:
        assert nPoolUpload(j, pool.ctypes.data, pool.size) == 0 and nResize(j, w, h) == 0
        xy, waited = np.zeros(2, np.int32), np.zeros(1, np.int64)
        assert nPickInfo(j, xy.ctypes.data, waited.ctypes.data) == 0 and tuple(xy) == (w // 2, h // 2)
        picks = []
        for i in range(n):
            assert nSetCamera(j, *[float(v) for v in cams[i]]) == 0
            assert nSetParams(j, int(fns[i]), 0, int(pool.size), 0, 2, 0, 1) == 0
            assert nDispatchAsync(j) == 0
            d = np.zeros(1, np.float32)
            assert nReadPixel(j, w // 2, h // 2, 0, d.ctypes.data, 0) == 0
            picks.append(d.view(np.uint32)[0])
        assert nPickInfo(j, 0, waited.ctypes.data) == n and waited[0] == 0
        full = np.zeros((h, w), np.float32)
        assert nReadDepth(j, full.ctypes.data) == 0        # the last dispatched frame, whole
        assert full.view(np.uint32)[h // 2, w // 2] == picks[-1]
        for i in range(0, n, 3):
            ref = oracle.render(pool, w, h, cams[i], int(fns[i]), 0, rows=(h // 2, h // 2 + 1), want_hits=False)
            assert ref["depth"].view(np.uint32)[h // 2, w // 2] == picks[i], i
        assert nSetPick(j, -1, -1) == 0 and nSync(j) == 0
    finally:
        assert nDestroy(j) == 0


def test_what_an_enqueued_dispatch_leaves_for_a_host_that_does_not_read_back(ctx):
    """dispatchCompute of both host mirrors enqueues (svo_dispatch_async).  A host that looks at svo_get_stats or at the
    device images right behind it, without a read-back, sees the frame only after svo_sync: last_dispatch_ms is the GPU time of
    the last WAITING dispatch (svo_dispatch), never of an enqueued one, and svo_output_device_ptrs names the images of the last
    dispatched frame -- which take turns on four sets while the library owns them."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(256)
    w, h = 256, 144
    ctx.pool_upload(pool)
    ctx.resize(w, h)
    ctx.set_camera(CAMERAS["K1"])
    ctx.set_params(2, 0, 0, 0, 2, 0, 1)
    ctx.dispatch()
    t_wait = ctx.stats()["last_dispatch_ms"]
    assert t_wait > 0
    want = {}
    for f in (3, 4, 5):
        ctx.set_params(f, 0, 0, 0, 2, 0, 1)
        ctx.dispatch()
        want[f] = ctx.read_color()
    t_wait = ctx.stats()["last_dispatch_ms"]
    seen = []
    for f in (3, 4, 5, 3, 4):
        ctx.set_params(f, 0, 0, 0, 2, 0, 1)
        ctx.dispatch_async()
        assert ctx.stats()["last_dispatch_ms"] == t_wait          # not this frame's time: nothing waited for it
        col_ptr = ctx.output_device_ptrs()[0]
        seen.append(col_ptr)
        ctx.sync()                                                # ... after which the device image is the frame
        dev = ctx.dev_read(col_ptr, w * h * 4).reshape(h, w, 4)
        assert (dev == want[f]).all(), f
    assert len(set(seen)) == 4 and seen[0] == seen[4]             # four image sets in turn (the default) ...
    ctx.set_overlap(2)                                            # ... or as many as asked for
    two = []
    for f in (3, 4, 5, 3):
        ctx.set_params(f, 0, 0, 0, 2, 0, 1)
        ctx.dispatch_async()
        two.append(ctx.output_device_ptrs()[0])
    ctx.sync()
    assert len(set(two)) == 2 and two[0] == two[2] and two[1] == two[3]
    with pytest.raises(Exception):
        ctx.set_overlap(9)
    ctx.set_overlap(False)
    ptrs = set()
    for f in (3, 4):
        ctx.set_params(f, 0, 0, 0, 2, 0, 1)
        ctx.dispatch_async()
        ptrs.add(ctx.output_device_ptrs()[0])
    ctx.sync()
    assert len(ptrs) == 1                                         # one set, as before round 6


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_mixes_of_overlapped_dispatches_picks_and_state_changes(seed):
    """The alternation's bookkeeping under a random host: number of image sets, pick position, pipeline, image size, camera,
    render mode and frame number change between dispatches; some frames are read back whole, of some only the pick, some not at
    all; svo_sync and waiting dispatches come in between.  Every value read must be what a second context renders synchronously
    (one stream, one image set, no pick) for the same state."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    rng = np.random.RandomState(seed)
    pool, _ = scene.build_scene3(256, 1, 8, 96)
    ctx, ref = hiplib.HipContext(0), hiplib.HipContext(0)
    try:
        sizes = [(208, 120), (131, 77)]
        w, h = sizes[0]
        for c in (ctx, ref):
            c.pool_upload(pool)
            c.resize(w, h)
        ref.set_overlap(0)
        ref.set_pick(-1, -1)
        cams = [CAMERAS["K0"], CAMERAS["K1"], CAMERAS["K2"]]
        cache = {}

        def want(key):
            if key not in cache:
                (ww, hh), ci, frame, mode, pipe = key
                ref.resize(ww, hh)
                ref.set_pipeline(pipe)
                cache[key] = ref.render(None, None, None, cams[ci], frame, mode)
            return cache[key]

        pick = (w // 2, h // 2)
        frame, last = 2, None
        for step in range(60):
            r = rng.rand()
            if r < 0.12:
                ctx.set_overlap(int(rng.choice([0, 1, 2, 3, 5, 8])))
            elif r < 0.24:
                if rng.rand() < 0.25:
                    ctx.set_pick(-1, -1)
                    pick = None
                else:
                    pick = (int(rng.randint(0, w)), int(rng.randint(0, h)))
                    ctx.set_pick(*pick)
            elif r < 0.30:
                w, h = sizes[int(rng.randint(0, 2))]
                ctx.resize(w, h)
                if pick is not None and (pick[0] >= w or pick[1] >= h):
                    pick = None                               # (svo_resize drops a pick that fell outside the new image)
                    assert ctx.pick_info()["x"] == -1
            elif r < 0.36:
                ctx.sync()
            pipe = int(rng.choice([1, 1, 1, 0]))
            ctx.set_pipeline(pipe)
            ci, mode = int(rng.randint(0, 3)), int(rng.choice([0, 0, 2, 3]))
            ctx.set_camera(cams[ci])
            ctx.set_params(frame, mode, 0, 0, 2, 0, 1)
            key = ((w, h), ci, frame, mode, pipe)
            if rng.rand() < 0.15:
                ctx.dispatch()
            else:
                ctx.dispatch_async()
            frame += int(rng.randint(0, 3))
            tag = (seed, step, key)
            what = rng.rand()
            if what < 0.45 and pick is not None:
                i0 = ctx.pick_info()
                px = ctx.read_pixel(*pick)
                i1 = ctx.pick_info()
                assert i1["from_mail"] == i0["from_mail"] + 1, tag       # whole frames, one sample: every such frame has its pick launch
                _pixel_matches(px, want(key), pick[0], pick[1])
            elif what < 0.75:
                got, exp = _frame(ctx), want(key)
                assert (got["rgba"] == exp["rgba"]).all(), tag
                assert (got["depth"].view(np.uint32) == exp["depth"].view(np.uint32)).all(), tag
                assert (got["hits"] == exp["hits"]).all(), tag
            elif what < 0.85:
                x, y = int(rng.randint(0, w)), int(rng.randint(0, h))
                _pixel_matches(ctx.read_pixel(x, y), want(key), x, y)
        ctx.sync()
    finally:
        ctx.close()
        ref.close()
