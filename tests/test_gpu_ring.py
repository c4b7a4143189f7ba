"""svo_ring_*: frames in flight behind the C ABI (library-owned streams and images).  Every frame a ring renders is
the frame a dispatch of its own renders; partial batches, slot re-use, caller-owned slots, stripes, errors."""
import numpy as np
import pytest
import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    c = helpers.DualContext()
    pool, _ = scene.build_scene(256)
    c.pool_upload(pool)
    c.set_camera(CAMERAS["K1"])
    yield c
    c.close()


def _alone(ctx, w, h, frame, mode, **kw):
    return ctx.render(None, w, h, None, frame, mode, **kw)


def _eq(a, b, hits=True):
    assert (a["rgba"] == b["rgba"]).all()
    assert (a["depth"].view(np.uint32) == b["depth"].view(np.uint32)).all()
    if hits:
        assert a["hits"].tobytes() == b["hits"].tobytes()


@pytest.mark.parametrize("pipeline", [0, 1, 2])
@pytest.mark.parametrize("slots,per", [(1, 1), (3, 4), (4, 5)])
def test_ring_frames_equal_frames_dispatched_alone(ctx, pipeline, slots, per):
    w, h = 200, 120
    ctx.set_pipeline(pipeline)
    ctx.set_tuning(10 if slots > 1 else 0, 9)
    ctx.resize(w, h)
    want = {f: _alone(ctx, w, h, f, 0) for f in range(2, 2 + 2 * slots * per + 3)}
    ctx.set_params(2, 0, 0, 0, 2, 0, 1)
    ctx.ring_create(slots, per, want_hits=True)
    frame, held = 2, {}
    for rnd in range(2):                                  # the second round re-uses every slot
        for _ in range(slots):
            n = per if not (rnd == 1 and _ == slots - 1) else max(1, per - 1)   # a partial batch at the end
            s = ctx.ring_submit(frame, n)
            held[s] = (frame, n)
            frame += n
    for s, (first, n) in held.items():
        q = ctx.ring_query(s)
        ctx.ring_wait(s)
        q = ctx.ring_query(s)
        assert q["done"] and q["first_frame"] == first and q["nframes"] == n and q["gpu_ms"] > 0
        for k in range(n):
            _eq(ctx.ring_read(s, k, want_hits=True), want[first + k])
        with pytest.raises(Exception):
            ctx.ring_read(s, n)
    # the context's own images and stream are untouched by the ring
    _eq(_alone(ctx, w, h, 3, 0), want[3])
    ctx.ring_destroy()
    ctx.set_tuning(0, 0)


def test_ring_modes_beam_and_path_options(ctx):
    from oracle import oracle
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(256)
    w, h = 160, 96
    ctx.set_pipeline(1)
    ctx.resize(w, h)
    ctx.ring_create(2, 3)
    for mode, kw in ((2, {}), (0, dict(bounces=3, mirror_mask=0b1000)), (0, dict(use_beam=1)), (1, {}), (0, dict(spp=3))):
        ctx.set_params(5, mode, 0, kw.get("use_beam", 0), kw.get("bounces", 2), kw.get("mirror_mask", 0), kw.get("spp", 1))
        s = ctx.ring_submit(5, 3)
        for k in (0, 2):
            got = ctx.ring_read(s, k)
            ref = oracle.render(pool, w, h, CAMERAS["K1"], 5 + k, mode, bounces=kw.get("bounces", 2),
                                mirror_mask=kw.get("mirror_mask", 0), spp=kw.get("spp", 1), want_hits=False)
            _eq(got, ref, hits=False)
    ctx.ring_destroy()


def test_ring_errors_and_lifetime(ctx):
    from svo_raytracer_amd.hiplib import SvoError
    ctx.resize(64, 64)
    with pytest.raises(SvoError):
        ctx.ring_submit(2, 1)             # no ring
    with pytest.raises(SvoError):
        ctx.ring_create(0, 1)
    with pytest.raises(SvoError):
        ctx.ring_create(9, 1)
    ctx.ring_create(2, 2)
    with pytest.raises(SvoError):
        ctx.ring_submit(2, 3)             # more than a slot holds
    with pytest.raises(SvoError):
        ctx.ring_wait(2)
    with pytest.raises(SvoError):
        ctx.ring_read(0, 0)               # nothing submitted to the slot yet
    s = ctx.ring_submit(2, 2)
    ctx.resize(80, 64)                    # a new image size takes the ring with it (after the frames in flight)
    with pytest.raises(SvoError):
        ctx.ring_submit(4, 1)
    ctx.ring_create(1, 1)
    ctx.ring_destroy()
    ctx.ring_destroy()                    # idempotent
    # svo_set_stream(NULL) returns to the library's own stream
    ctx.set_stream(None)
    ctx.set_params(2, 2)
    ctx.dispatch()


def test_ring_with_caller_owned_slots_and_stripes(ctx):
    """What the multi-GPU frame ring does: every slot renders one rank's packed stripes into a chunk of a gather buffer."""
    import torch
    from svo_raytracer_amd.tiles import stripe_layout
    w, h, world, rank, per = 200, 120, 3, 1, 2
    ctx.set_pipeline(1)
    ctx.resize(w, h)
    full = {f: _alone(ctx, w, h, f, 0) for f in (2, 3)}
    first, step, n, _, rpr = stripe_layout(h, world, rank)
    ctx.set_stripes(first, step, n, 0)
    ctx.set_params(2, 0, 0, 0, 2, 0, 1)
    ctx.ring_create(2, per)
    buf = torch.zeros((2, per, rpr, w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    stride = rpr * w
    ctx.ring_bind_slot(0, buf[0].data_ptr(), buf[1].data_ptr(), None, stride)
    s = ctx.ring_submit(2, per)
    assert s == 0
    ctx.ring_wait(s)
    torch.cuda.synchronize()
    col, dep = buf[0].cpu().numpy(), buf[1].cpu().numpy().view(np.float32)
    for k in range(per):
        for jrow in range(n):
            y0 = (first + jrow * step) * 8
            rows = min(8, h - y0)
            if rows <= 0:
                continue
            assert (col[k, jrow * 8:jrow * 8 + rows].view(np.uint8).reshape(rows, w, 4) == full[2 + k]["rgba"][y0:y0 + rows]).all()
            assert (dep[k, jrow * 8:jrow * 8 + rows].view(np.uint32) == full[2 + k]["depth"][y0:y0 + rows].view(np.uint32)).all()
    p = ctx.ring_device_ptrs(0)
    assert p["color"] == buf[0].data_ptr() and p["frame_stride"] == stride and p["stream"]
    ctx.ring_bind_slot(0, None, None, None, 0)
    assert ctx.ring_device_ptrs(0)["frame_stride"] == w * h
    ctx.ring_destroy()
    ctx.resize(w + 8, h)   # resets the stripes


def test_batch_stride_must_cover_a_frame(ctx):
    import torch
    from svo_raytracer_amd.hiplib import SvoError
    w, h = 128, 80
    ctx.set_pipeline(1)
    ctx.resize(w, h)
    buf = torch.zeros((2, 3, h, w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.bind_outputs(buf[0].data_ptr(), buf[1].data_ptr(), None)
    ctx.set_params(2, 0)
    ctx.set_batch(3, w * h - 1)          # one element short: frames would overlap
    with pytest.raises(SvoError):
        ctx.dispatch()
    ctx.set_batch(3, w * h)
    ctx.dispatch()
    ctx.set_batch(1, 0)
    ctx.bind_outputs(None, None, None)


def test_pool_written_through_the_device_pointer_needs_commit(ctx):
    """svo_pool_reserve + svo_pool_device_ptr + (the caller's copy, e.g. an RCCL broadcast) + svo_pool_commit"""
    import ctypes
    import torch
    import svo_raytracer_amd.scene as scene
    from oracle import oracle
    from svo_raytracer_amd.cameras import CAMERAS
    a, _ = scene.build_scene(128)
    b, _ = scene.build_scene(64)
    w, h = 96, 64
    for pipeline in (0, 1):
        ctx.set_pipeline(pipeline)
        ctx.render(a, w, h, CAMERAS["K1"], 2, 0)           # the context holds pool a (and its derived tables)
        ctx.pool_reserve(b.size)
        ptr, n = ctx.pool_device_ptr()
        assert n == b.size
        src = torch.from_numpy(b).cuda()
        torch.cuda.synchronize()
        # the caller's own asynchronous copy into the library's pool, on torch's stream, as a broadcast would be
        hip = ctypes.CDLL("libamdhip64.so.7")   # the soname torch's runtime is already mapped under: the same library, not a second one
        hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        assert hip.hipMemcpyAsync(ctypes.c_void_p(ptr), ctypes.c_void_p(src.data_ptr()), b.size, 3,
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        ctx.pool_commit()
        got = ctx.render(None, w, h, CAMERAS["K1"], 2, 0)
        ref = oracle.render(b, w, h, CAMERAS["K1"], 2, 0)
        assert (got["rgba"] == ref["rgba"]).all() and (got["hits"]["pointer"] == ref["hits"]["pointer"]).all()
    ctx.pool_upload(scene.build_scene(256)[0])              # what the module's other tests expect


def test_ring_streams_with_reserved_cus_and_device_memory_calls(ctx):
    """svo_set_reserved_cus: the next ring's streams carry a CU mask (one CU per XCD left free); same frames.
    svo_dev_alloc / _read / _free and the IPC calls' argument checks."""
    from svo_raytracer_amd.hiplib import SvoError
    w, h = 160, 96
    ctx.set_pipeline(1)
    ctx.resize(w, h)
    want = {f: _alone(ctx, w, h, f, 0) for f in range(2, 8)}
    ctx.set_params(2, 0, 0, 0, 2, 0, 1)
    for reserve in (1, 2, 0):
        ctx.set_reserved_cus(reserve)
        ctx.ring_create(2, 3, want_hits=True)
        s0, s1 = ctx.ring_submit(2, 3), ctx.ring_submit(5, 3)
        for s, first in ((s0, 2), (s1, 5)):
            for k in range(3):
                _eq(ctx.ring_read(s, k, want_hits=True), want[first + k])
        ctx.ring_destroy()
    with pytest.raises(SvoError):
        ctx.set_reserved_cus(17)
    p = ctx.dev_alloc(4096)
    assert (ctx.dev_read(p, 4096) == 0).all()        # zeroed
    handle = ctx.ipc_export(p)
    assert len(handle) == 64
    with pytest.raises(SvoError):
        ctx.dev_free(p + 8)                           # not an allocation of this context
    ctx.dev_free(p)
    with pytest.raises(SvoError):
        ctx.dev_free(p)                               # already freed
    with pytest.raises(SvoError):
        ctx.ipc_close(12345)


@pytest.mark.parametrize("w,h,spp,per", [(1920, 1080, 8, 5), (1920, 1080, 16, 2), (3840, 2160, 1, 17)])
def test_large_batches_where_the_refill_reciprocal_is_inexact(ctx, w, h, spp, per):
    """The refill turns a slot number into (frame of the batch, tile) with a multiply-high by ceil(2^32 / d); that is
    exact only while e (n + d) < 2^32 (svo_persistent.hip.h::udiv_magic).  Full-HD with 8 samples x 5 frames, 16 samples
    x 2 frames and a 4K batch of 17 frames are past it (wrong frame index for the last tiles: dropped or misplaced
    tiles); the launch then divides for real.  Every frame of the batch = the frame dispatched alone."""
    ctx.set_pipeline(1)
    ctx.set_tuning(10, 9)
    ctx.resize(w, h)
    picks = sorted({0, per // 2, per - 1})
    want = {k: _alone(ctx, w, h, 2 + k, 0, spp=spp) for k in picks}
    ctx.set_params(2, 0, 0, 0, 2, 0, spp)
    ctx.ring_create(1, per, want_hits=False)
    s = ctx.ring_submit(2, per)
    ctx.ring_wait(s)
    for k in picks:
        _eq(ctx.ring_read(s, k), want[k], hits=False)
    ctx.ring_destroy()
    ctx.set_tuning(0, 0)


@pytest.mark.parametrize("pipeline", [1, 0, 2])
def test_frames_with_their_own_cameras_in_flight(ctx, pipeline):
    """svo_ring_submit_cams: 24 distinct cameras in flight (6 slots x 4 frames, one persistent launch per slot), the
    camera path of a user who moves (Camera.rotate + strafe through the host mirror; frameNumber reset to 1 by the motion,
    Main.java:225-233, 275) plus frames at rest in between (frameNumber counting on).  Every frame = svo_set_camera +
    svo_dispatch of its own.  Through the JNI-typed export on the persistent pipeline."""
    import ctypes
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS, orbit_path
    w, h = 256, 144
    cams, fns = orbit_path(24, yaw_step=0.02, pitch_step=0.003, forward=0.002, side=0.001)
    fns = fns.copy()
    cams[10:13] = cams[9]; fns[10:13] = [2, 3, 4]          # the user pauses for three frames
    cams[23] = CAMERAS["K0"]; fns[23] = 77                  # and a cut to another camera
    ctx.set_pipeline(pipeline)
    ctx.set_tuning(10 if pipeline == 1 else 0, 9)
    ctx.resize(w, h)
    want = []
    for k in range(24):
        ctx.set_camera(cams[k])
        want.append(_alone(ctx, w, h, int(fns[k]), 0))
    ctx.set_camera(CAMERAS["K2"])                           # the context's own camera is not what the frames use ...
    ctx.set_params(5, 0, 0, 0, 2, 0, 1)
    ctx.ring_create(6, 4, want_hits=True)
    slots = []
    for b in range(6):
        if pipeline == 1 and b % 2 == 0:
            L = hiplib.lib()
            fn = L.Java_src_engine_HipRenderer_nRingSubmitCams
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64]
            fn.restype = ctypes.c_int32
            cc = cams[4 * b:4 * b + 4].copy(); ff = fns[4 * b:4 * b + 4].copy()
            slots.append(fn(None, None, ctx._h.value, 4, cc.ctypes.data, ff.ctypes.data))
            cc[:] = 0; ff[:] = 0                            # the arrays were copied before the call returned
        else:
            slots.append(ctx.ring_submit_cams(cams[4 * b:4 * b + 4], fns[4 * b:4 * b + 4]))
    assert slots == list(range(6))
    for b in range(6):
        ctx.ring_wait(b)
        q = ctx.ring_query(b)
        assert q["nframes"] == 4 and q["first_frame"] == int(fns[4 * b])
        for k in range(4):
            _eq(ctx.ring_read(b, k, want_hits=True), want[4 * b + k])
    # ... and is still there afterwards, as is its frameNumber
    _eq(_alone(ctx, w, h, 5, 0), ctx.render(None, w, h, CAMERAS["K2"], 5, 0))
    # one frame per submission takes the plain path (beam pre-pass allowed), a batch with the beam is refused
    ctx.set_params(5, 0, 0, 1, 2, 0, 1)
    s = ctx.ring_submit_cams(cams[3:4], fns[3:4])
    _eq(ctx.ring_read(s, 0, want_hits=False), want[3], hits=False)
    with pytest.raises(Exception):
        ctx.ring_submit_cams(cams[:2], fns[:2])
    ctx.set_params(5, 0, 0, 0, 2, 0, 1)
    ctx.ring_destroy()
    ctx.set_tuning(0, 0)
    ctx.set_camera(CAMERAS["K1"])
    ctx.set_pipeline(1)


def test_hostile_cameras_as_frames_of_one_launch_match_the_reference_shader():
    """The fuzz fixture's cases (tests/golden/fuzz_golden.npz: NaN / infinite / denormal / 1e30 camera components, cameras on
    faces, corners and cell boundaries, frame numbers up to +-2^31, every render mode) grouped by pool, image size and mode and
    rendered as the frames of svo_ring_submit_cams launches -- up to eight different hostile cameras and frame numbers in ONE
    persistent launch (the `cams` kernels) -- against the reference shader's own images."""
    import os
    from helpers import compare_with_golden
    from svo_raytracer_amd import hiplib
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_golden.npz"))
    groups = {}
    for s in z["index"]:
        name, pk = s.split(":")
        w, h, frame, mode, same = (int(v) for v in z[name + "/meta"])
        path = tuple(int(v) for v in z[name + "/path"]) if name + "/path" in z.files else (2, 0)
        groups.setdefault((pk, w, h, mode, path), []).append((name, frame))
    c = helpers.DualContext()
    ncases = nlaunch = 0
    try:
        c.set_pipeline(1)
        for (pk, w, h, mode, (bounces, mirror)), cases in sorted(groups.items()):
            if len(cases) < 2:
                continue
            c.pool_upload(z["pool/" + pk])
            c.resize(w, h)
            c.set_params(2, mode, 0, 0, bounces, mirror, 1)
            c.ring_create(1, 8, want_hits=True)
            for i in range(0, len(cases), 8):
                part = cases[i:i + 8]
                cams = np.stack([z[n + "/cam"] for n, _ in part]).astype(np.float32)
                s = c.ring_submit_cams(cams, [f for _, f in part])
                for k, (name, frame) in enumerate(part):
                    g = dict(rgba=z[name + "/rgba"], depth_bits=z[name + "/depth_bits"], first_hit=z[name + "/first_hit"])
                    bad = compare_with_golden(c.ring_read(s, k, want_hits=True), g)
                    assert bad == {k2: 0 for k2 in bad}, (name, pk, mode, bad)
                    ncases += 1
                nlaunch += 1
            c.ring_destroy()
    finally:
        c.close()
    assert ncases >= 300 and nlaunch < ncases / 2


def test_a_ring_of_several_slots_runs_the_benchmarked_launch_shape_by_itself(ctx):
    """What a drop-in host gets without any tuning call: a dispatch on the context's stream fills the GPU, the
    submissions of a ring with more than one slot take 10 persistent waves per CU -- the shape bench.py's headline is
    measured on; a positive svo_set_tuning value is used as given, 0 returns to the automatic choice.  Same bytes in every shape."""
    import ctypes
    from svo_raytracer_amd import hiplib
    w, h = 1920, 1080                      # 32 400 tiles: more work than any launch shape has waves
    ctx.set_pipeline(1)
    ctx.set_tuning(0, 0)
    ctx.resize(w, h)
    want = _alone(ctx, w, h, 4, 0)
    fill = ctx.launch_info()
    assert fill["waves_per_cu"] >= 16 and fill["waves"] == fill["waves_per_cu"] * 256 and fill["round_threshold_sixteenths"] == 9
    ctx.set_params(4, 0, 0, 0, 2, 0, 1)
    ctx.ring_create(1, 2, want_hits=True)            # one slot = one launch at a time: fill
    s = ctx.ring_submit(4, 2)
    _eq(ctx.ring_read(s, 0, want_hits=True), want)
    assert ctx.launch_info() == fill
    ctx.ring_create(3, 2, want_hits=True)            # several slots: the benchmarked shape
    s = ctx.ring_submit(4, 2)
    _eq(ctx.ring_read(s, 0, want_hits=True), want)
    assert ctx.launch_info() == {"waves": 2560, "waves_per_cu": 10, "round_threshold_sixteenths": 9}
    _eq(_alone(ctx, w, h, 4, 0), want)              # a dispatch next to the ring still fills the GPU
    assert ctx.launch_info() == fill
    ctx.set_tuning(7, 11)                            # the caller's shape, everywhere
    s = ctx.ring_submit(4, 1)
    _eq(ctx.ring_read(s, 0, want_hits=True), want)
    assert ctx.launch_info() == {"waves": 7 * 256, "waves_per_cu": 7, "round_threshold_sixteenths": 11}
    ctx.set_tuning(0, 0)
    s = ctx.ring_submit(4, 1)
    ctx.ring_wait(s)
    assert ctx.launch_info() == {"waves": 2560, "waves_per_cu": 10, "round_threshold_sixteenths": 9}
    # the same through the JNI-typed export HipRenderer.lastLaunchWaves binds
    f = getattr(hiplib.lib(), "Java_src_engine_HipRenderer_nLaunchInfo")
    f.restype = ctypes.c_int32
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    wpc = ctypes.c_int32(0)
    assert f(None, None, ctx._h.value, ctypes.addressof(wpc)) == 2560 and wpc.value == 10
    ctx.ring_destroy()


def test_a_continued_accumulation_is_refused_on_a_ring_of_several_slots(ctx):
    """svotrace.comp:712-719 blends with the image the PREVIOUS dispatch left; on a ring of several slots that image is in
    another slot: refused with a message instead of a silently different recurrence.  Fresh sequences and rings
    of one slot go through."""
    ctx.set_pipeline(1)
    ctx.resize(160, 96)
    ctx.set_params(2, 0, 0, 0, 2, 0, 1)
    ctx.set_progressive(True)
    try:
        ctx.ring_create(2, 1)
        ctx.set_sequence(1, False)
        with pytest.raises(Exception, match="ONE slot"):
            ctx.ring_submit(2, 1)
        ctx.set_sequence(3, False)
        with pytest.raises(Exception, match="ONE slot"):
            ctx.ring_submit(2, 1)
        ctx.set_sequence(3, True)
        ctx.ring_wait(ctx.ring_submit(2, 1))
        ctx.ring_create(1, 1)
        ctx.set_sequence(2, False)
        ctx.ring_wait(ctx.ring_submit(5, 1))
    finally:
        ctx.set_progressive(False)
        ctx.set_sequence(1, False)
        ctx.ring_destroy()
