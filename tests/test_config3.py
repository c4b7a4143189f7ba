"""BASELINE.json config 3 -- what bench.py measures: 8192^3 procedural SVO, 1920x1080, primary + 1 bounce -- the 4K /
5-segment / mirror frame of config 4, and config 2 (2048^3, primary rays only), against the reference shader's own output at full size (llvmpipe golden of every
8th pixel in x and y: tests/golden/make_golden_config3.py).  CPU leg: the oracle on the same pixels.  GPU leg: the HIP
pipelines' full frames, sampled; for the persistent pipeline also as bench.py runs it (frames in flight, batched)."""
import os
import zlib

import numpy as np
import pytest

import poolcache

import svo_raytracer_amd.scene as scene
import helpers

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config3_8192.npz")


def _cases():
    return [str(n) for n in np.load(GOLD)["index"]]


@pytest.fixture(scope="module")
def pool8192():
    z = np.load(GOLD)
    pool = poolcache.pool()
    assert pool.size == int(z["pool_size"][0]) and zlib.crc32(pool.tobytes()) == int(z["pool_crc32"][0]), \
        "scene generator drifted: regenerate tests/golden/config3_8192.npz"
    return pool


@pytest.fixture(scope="module")
def pool2048():
    z = np.load(GOLD)
    pool, _ = scene.build_scene(2048)
    assert pool.size == int(z["2048_pool_size"][0]) and zlib.crc32(pool.tobytes()) == int(z["2048_pool_crc32"][0]), \
        "scene generator drifted: regenerate tests/golden/config3_8192.npz"
    return pool


def _meta(z, name):
    w, h, frame, mode, bounces, mirror = (int(v) for v in z[name + "/meta"])
    return w, h, frame, mode, bounces, (0xfffffffd if mirror else 0)


def _check(res, z, name, step, sampled=False):
    """res: full-size images (sampled=False) or images that only hold every step-th pixel's value at its place"""
    sub = (slice(0, res["rgba"].shape[0], step), slice(0, res["rgba"].shape[1], step))
    fh = z[name + "/first_hit"]
    hit = fh[..., 0] != 0
    assert (res["rgba"][sub] == z[name + "/rgba"]).all(), name
    assert (res["depth"].view(np.uint32)[sub] == z[name + "/depth_bits"]).all(), name
    h = res["hits"]
    assert (h["pointer"][sub] == fh[..., 0]).all(), name                 # hit voxel IDs
    assert ((h["value"][sub] == fh[..., 1]) | ~hit).all(), name
    assert ((h["raw_normal"][sub] == fh[..., 2]) | ~hit).all(), name     # packed normals
    assert ((h["level"][sub] == (fh[..., 3] >> 16)) | ~hit).all(), name
    assert ((h["iter"][sub] == (fh[..., 3] & 0xFFFF)) | ~hit).all(), name


@pytest.mark.parametrize("name", _cases())
def test_config3_oracle_matches_reference_at_full_size(pool8192, name):
    from oracle import oracle
    z = np.load(GOLD)
    step = int(z["step"][0])
    w, h, frame, mode, bounces, mirror = _meta(z, name)
    res = oracle.render(pool8192, w, h, z[name + "/cam"], frame, mode, bounces=bounces, mirror_mask=mirror, xstep=step, ystep=step)
    _check(res, z, name, step)


def test_config2_oracle_matches_reference_at_full_size(pool2048):
    from oracle import oracle
    z = np.load(GOLD)
    step = int(z["step"][0])
    for name in (str(n) for n in z["index2048"]):
        w, h, frame, mode, bounces, mirror = _meta(z, name)
        res = oracle.render(pool2048, w, h, z[name + "/cam"], frame, mode, xstep=step, ystep=step)
        _check(res, z, name, step)


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_config2_hip_matches_reference_at_full_size(pool2048, pipeline):
    from svo_raytracer_amd import hiplib
    z = np.load(GOLD)
    step = int(z["step"][0])
    ctx = helpers.DualContext()
    try:
        ctx.set_pipeline(pipeline)
        ctx.pool_upload(pool2048)
        for name in (str(n) for n in z["index2048"]):
            w, h, frame, mode, bounces, mirror = _meta(z, name)
            _check(ctx.render(None, w, h, z[name + "/cam"], frame, mode), z, name, step)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_config3_hip_matches_reference_at_full_size(pool8192, pipeline):
    from svo_raytracer_amd import hiplib
    z = np.load(GOLD)
    step = int(z["step"][0])
    ctx = helpers.DualContext()
    try:
        ctx.set_pipeline(pipeline)
        ctx.pool_upload(pool8192)
        for name in _cases():
            w, h, frame, mode, bounces, mirror = _meta(z, name)
            res = ctx.render(None, w, h, z[name + "/cam"], frame, mode, bounces=bounces, mirror_mask=mirror)
            _check(res, z, name, step)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_config3_as_the_benchmark_runs_it(pool8192):
    """frames 2 and 57 of the golden inside bench.py's throughput configuration: svo_set_tuning(10, 9), 4 dispatches of 5
    frames in flight on 4 streams (frames 2..21 and 42..61), the two golden frames picked out of their batches"""
    import torch
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    z = np.load(GOLD)
    step = int(z["step"][0])
    w, h = 1920, 1080
    ctx = helpers.DualContext()
    try:
        ctx.set_pipeline(1)
        ctx.pool_upload(pool8192)
        ctx.resize(w, h)
        ctx.set_camera(CAMERAS["K1"])
        ctx.set_tuning(10, 9)
        nd, nb = 4, 5
        for first, want in ((2, ("c3_f2", 2)), (42, ("c3_f57", 57))):
            col = [torch.zeros((nb, h, w), dtype=torch.int32, device="cuda") for _ in range(nd)]
            dep = [torch.zeros((nb, h, w), dtype=torch.float32, device="cuda") for _ in range(nd)]
            hit = [torch.zeros((nb, h, w, 4), dtype=torch.int32, device="cuda") for _ in range(nd)]
            streams = [torch.cuda.Stream() for _ in range(nd)]
            torch.cuda.synchronize()
            ctx.set_batch(nb, w * h)
            for b in range(nd):
                ctx.set_stream(streams[b].cuda_stream)
                ctx.bind_outputs(col[b].data_ptr(), dep[b].data_ptr(), hit[b].data_ptr())
                ctx.set_params(first + b * nb, 0, 0, 0, 2, 0, 1)
                ctx.dispatch_async()
            torch.cuda.synchronize()
            b, k = divmod(want[1] - first, nb)
            res = {"rgba": col[b][k].cpu().numpy().view(np.uint8).reshape(h, w, 4), "depth": dep[b][k].cpu().numpy(),
                   "hits": hit[b][k].cpu().numpy().reshape(-1, 4).copy().view(hiplib.HIT_DTYPE).reshape(h, w)}
            _check(res, z, want[0], step)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.bind_outputs(None, None, None)
        ctx.set_batch(1, 0)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_config3_through_the_jni_ring(pool8192):
    """The same throughput configuration behind the drop-in boundary: JNI-typed calls only (what HipRenderer.java's
    createFrameRing / submitFrames / readFrame make) -- a library-owned ring of 6 slots x 4 frames, six submissions in
    flight (bench.py's default shape), the golden frames 2 and 57 read back out of their slots; no torch stream or tensor involved,
    and no tuning / pipeline call: the library's defaults are the benchmarked configuration."""
    import ctypes
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

    nCreate, nDestroy = fn("nCreate", jlong, jint), fn("nDestroy", jint, jlong)
    nPoolUpload = fn("nPoolUpload", jint, jlong, jlong, jlong)
    nSetCamera = fn("nSetCamera", jint, jlong, *([jfloat] * 15))
    nSetParams = fn("nSetParams", jint, jlong, *([jint] * 7))
    nResize = fn("nResize", jint, jlong, jint, jint)
    nLaunchInfo = fn("nLaunchInfo", jint, jlong, jlong)
    nRingCreate = fn("nRingCreate", jint, jlong, jint, jint, jint)
    nRingSubmit = fn("nRingSubmit", jint, jlong, jint, jint)
    nRingWait, nRingDone = fn("nRingWait", jint, jlong, jint), fn("nRingDone", jint, jlong, jint, jlong)
    nRingReadColor = fn("nRingReadColor", jint, jlong, jint, jint, jlong)
    nRingReadDepth = fn("nRingReadDepth", jint, jlong, jint, jint, jlong)
    nRingReadHits = fn("nRingReadHits", jint, jlong, jint, jint, jlong)
    nRingReadPixel = fn("nRingReadPixel", jint, jlong, jint, jint, jint, jint, jlong, jlong, jlong)
    nRingDestroy = fn("nRingDestroy", jint, jlong)
    nDerivedInfo = fn("nDerivedInfo", jlong, jlong, jlong)

    z = np.load(GOLD)
    step = int(z["step"][0])
    w, h, nd, nb = 1920, 1080, 6, 4      # bench.py's default shape: 6 submissions in flight x 4 frames
    j = nCreate(0)
    assert j != 0
    try:
        assert nPoolUpload(j, pool8192.ctypes.data, pool8192.size) == 0
        # no nSetPipeline / nSetTuning call: a new context runs pipeline 1, and a ring of several slots takes the benchmarked
        # launch shape (10 persistent waves per CU, rounds at 9/16) by itself -- checked through nLaunchInfo below
        assert nSetCamera(j, *[float(v) for v in np.asarray(CAMERAS["K1"], np.float32).reshape(-1)]) == 0
        assert nRingSubmit(j, 2, 1) < 0                       # no ring yet
        assert nResize(j, w, h) == 0
        assert nRingCreate(j, nd, nb, 1) == 0
        walk = ctypes.c_int32(-1)
        assert nDerivedInfo(j, ctypes.addressof(walk)) > 1000 and walk.value == 1   # the bench scene takes the table walk
        assert nSetParams(j, 2, 0, int(pool8192.size), 0, 2, 0, 1) == 0
        assert nRingSubmit(j, 2, nb + 1) < 0                  # more frames than a slot holds
        for first, want in ((2, ("c3_f2", 2)), (42, ("c3_f57", 57))):
            slots = [nRingSubmit(j, first + b * nb, nb) for b in range(nd)]
            assert slots == list(range(nd)) or sorted(slots) == list(range(nd)), slots
            wpc = ctypes.c_int32(0)
            assert nLaunchInfo(j, ctypes.addressof(wpc)) == 2560 and wpc.value == 10
            b, k = divmod(want[1] - first, nb)
            ms = ctypes.c_float(0)
            assert nRingWait(j, slots[b]) == 0 and nRingDone(j, slots[b], ctypes.addressof(ms)) == 1 and ms.value > 0
            rgba = np.zeros((h, w, 4), np.uint8)
            depth = np.zeros((h, w), np.float32)
            hits = np.zeros((h, w), hiplib.HIT_DTYPE)
            assert nRingReadColor(j, slots[b], k, rgba.ctypes.data) == 0
            assert nRingReadDepth(j, slots[b], k, depth.ctypes.data) == 0
            assert nRingReadHits(j, slots[b], k, hits.ctypes.data) == 0
            _check({"rgba": rgba, "depth": depth, "hits": hits}, z, want[0], step)
            one_d = np.zeros(1, np.float32)                    # the crosshair pick on a frame of the ring
            assert nRingReadPixel(j, slots[b], k, 960, 540, 0, one_d.ctypes.data, 0) == 0
            assert one_d.view(np.uint32)[0] == depth.view(np.uint32)[540, 960]
            assert nRingReadColor(j, slots[b], nb, rgba.ctypes.data) < 0      # the slot holds frames 0..nb-1
            for s in slots:
                assert nRingWait(j, s) == 0
        assert nRingDestroy(j) == 0
    finally:
        assert nDestroy(j) == 0
