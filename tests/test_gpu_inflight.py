"""The configuration bench.py measures, under the oracle: pipeline 1, svo_set_tuning(10, 9), several persistent
launches in flight on alternating streams, frameNumber advancing every frame -- plus the ordering the library
guarantees for that usage (include/svo_hip.h): per-frame counter sets and sample accumulators are re-used only
after the frame that used them has finished, and pool edits wait for every frame in flight."""
import numpy as np
import pytest
import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from svo_raytracer_amd import hiplib
    c = helpers.DualContext()
    yield c
    c.close()


@pytest.fixture(scope="module")
def pool8192():
    import poolcache
    return poolcache.pool()


class Frames:
    """n output sets (colour, depth, hits) as torch tensors + s streams; frame k -> set k, stream k % s."""

    def __init__(self, ctx, w, h, n, s, rows=None):
        import torch
        self.torch = torch
        self.ctx, self.w, self.h, self.n = ctx, w, h, n
        rows = rows or h
        self.col = [torch.zeros((rows, w), dtype=torch.int32, device="cuda") for _ in range(n)]
        self.dep = [torch.zeros((rows, w), dtype=torch.float32, device="cuda") for _ in range(n)]
        self.hit = [torch.zeros((rows, w, 4), dtype=torch.int32, device="cuda") for _ in range(n)]
        self.streams = [torch.cuda.Stream() for _ in range(s)]
        torch.cuda.synchronize()

    def enqueue(self, k, frame, mode, bounces=2, mirror=0, spp=1):
        st = self.streams[k % len(self.streams)]
        self.ctx.set_stream(st.cuda_stream)
        self.ctx.bind_outputs(self.col[k].data_ptr(), self.dep[k].data_ptr(), self.hit[k].data_ptr())
        self.ctx.set_params(frame, mode, 0, 0, bounces, mirror, spp)
        self.ctx.dispatch_async()

    def done(self):
        self.torch.cuda.synchronize()
        self.ctx.set_stream(self.torch.cuda.current_stream().cuda_stream)
        self.ctx.bind_outputs(None, None, None)

    def get(self, k):
        from svo_raytracer_amd import hiplib
        h = self.hit[k].cpu().numpy().reshape(-1, 4).copy().view(hiplib.HIT_DTYPE).reshape(self.hit[k].shape[0], self.w)
        return {"rgba": self.col[k].cpu().numpy().view(np.uint8).reshape(-1, self.w, 4), "depth": self.dep[k].cpu().numpy(),
                "hits": h}


def _eq(a, b):
    return (np.array_equal(a["rgba"], b["rgba"]) and np.array_equal(a["depth"].view(np.uint32), b["depth"].view(np.uint32))
            and a["hits"].tobytes() == b["hits"].tobytes())


def test_bench_configuration_frames_in_flight_8192_1080p(ctx, pool8192):
    """8192^3, 1920x1080, mode 0: 9 frames (frameNumber 2..10) enqueued back to back on 3 streams with the bench
    tuning; every frame bit-equal to the same frame rendered alone, and to the oracle on a pixel subsample."""
    from oracle import oracle
    from svo_raytracer_amd.cameras import CAMERAS
    w, h, cam = 1920, 1080, CAMERAS["K1"]
    ctx.set_pipeline(1)
    ctx.pool_upload(pool8192)
    ctx.resize(w, h)
    ctx.set_camera(cam)
    ctx.set_tuning(10, 9)
    fr = Frames(ctx, w, h, 9, 3)
    try:
        for k in range(9):
            fr.enqueue(k, 2 + k, 0)
        fr.done()
        ctx.set_tuning(0, 0)
        step = 24
        sub = (slice(0, h, step), slice(0, w, step))
        for k in range(9):
            got = fr.get(k)
            alone = ctx.render(None, None, None, None, 2 + k, 0)
            assert _eq(got, alone), "frame %d in flight differs from the frame rendered alone" % (2 + k)
            ref = oracle.render(pool8192, w, h, cam, 2 + k, 0, xstep=step, ystep=step)
            assert (ref["rgba"][sub] == got["rgba"][sub]).all(), k
            assert (ref["depth"].view(np.uint32)[sub] == got["depth"].view(np.uint32)[sub]).all(), k
            for f in ("pointer", "value", "raw_normal", "level", "iter"):
                assert (ref["hits"][f][sub] == got["hits"][f][sub]).all(), (k, f)
        # consecutive frames really are different frames (bounce directions change with frameNumber)
        assert not np.array_equal(fr.get(0)["rgba"], fr.get(1)["rgba"])
        alone = [fr.get(k) for k in range(9)]
        # bench.py's default: 4 dispatches in flight, each ONE launch of 5 consecutive frames (svo_set_batch)
        import torch
        from svo_raytracer_amd import hiplib
        nd, nb = 4, 5
        col = [torch.zeros((nb, h, w), dtype=torch.int32, device="cuda") for _ in range(nd)]
        dep = [torch.zeros((nb, h, w), dtype=torch.float32, device="cuda") for _ in range(nd)]
        hit = [torch.zeros((nb, h, w, 4), dtype=torch.int32, device="cuda") for _ in range(nd)]
        streams = [torch.cuda.Stream() for _ in range(nd)]
        torch.cuda.synchronize()
        ctx.set_tuning(10, 9)
        ctx.set_batch(nb, w * h)
        for b in range(nd):
            ctx.set_stream(streams[b].cuda_stream)
            ctx.bind_outputs(col[b].data_ptr(), dep[b].data_ptr(), hit[b].data_ptr())
            ctx.set_params(2 + b * nb, 0, 0, 0, 2, 0, 1)
            ctx.dispatch_async()
        torch.cuda.synchronize()
        ctx.set_batch(1, 0)
        fr.done()                      # back to the library's own images, one frame at a time
        ctx.set_tuning(0, 0)
        for b in range(nd):
            for k in range(nb):
                f = b * nb + k
                got = {"rgba": col[b][k].cpu().numpy().view(np.uint8).reshape(h, w, 4), "depth": dep[b][k].cpu().numpy(),
                       "hits": hit[b][k].cpu().numpy().reshape(-1, 4).copy().view(hiplib.HIT_DTYPE).reshape(h, w)}
                want = alone[f] if f < 9 else ctx.render(None, None, None, None, 2 + f, 0)
                assert _eq(got, want), "frame %d of a batched launch differs from the frame rendered alone" % (2 + f)
    finally:
        ctx.set_batch(1, 0)
        fr.done()
        ctx.set_tuning(0, 0)


@pytest.mark.parametrize("pipeline", [1, 2])
def test_golden_case_with_frames_in_flight(ctx, pipeline):
    """A reference-shader golden (llvmpipe) rendered while other frames of the same scene are in flight."""
    from helpers import compare_with_golden, golden_case, golden_cases
    name, poolkey = [c for c in golden_cases() if golden_case(*c)["mode"] == 0][0]
    g = golden_case(name, poolkey)
    ctx.set_pipeline(pipeline)
    ctx.pool_upload(g["pool"])
    ctx.resize(g["w"], g["h"])
    ctx.set_camera(g["cam"])
    ctx.set_tuning(10, 9)
    fr = Frames(ctx, g["w"], g["h"], 9, 3)
    try:
        for k in range(9):
            fr.enqueue(k, g["frame"] + (k % 3), g["mode"])
        fr.done()
        for k in range(0, 9, 3):
            bad = compare_with_golden(fr.get(k), g)
            assert bad == {key: 0 for key in bad}, (k, bad)
        for k in range(3, 9):
            assert _eq(fr.get(k), fr.get(k - 3))
    finally:
        fr.done()
        ctx.set_tuning(0, 0)


@pytest.mark.parametrize("pipeline", [1, 2])
def test_more_frames_in_flight_than_the_library_ring_holds(ctx, pipeline):
    """24 frames on 6 streams: the ring of per-frame counter sets (8) / queue sets (4) wraps several times while
    earlier users are still running; the event the previous user recorded orders the re-use."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(1024)
    w, h = 640, 360
    ctx.set_pipeline(pipeline)
    ctx.pool_upload(pool)
    ctx.resize(w, h)
    ctx.set_camera(CAMERAS["K1"])
    ctx.set_tuning(3, 9)
    fr = Frames(ctx, w, h, 24, 6)
    try:
        for k in range(24):
            fr.enqueue(k, 2 + k % 4, (0, 2)[k % 2])
        fr.done()
        ctx.set_tuning(0, 0)
        ref = {}
        for k in range(24):
            key = (2 + k % 4, (0, 2)[k % 2])
            if key not in ref:
                ref[key] = ctx.render(None, None, None, None, key[0], key[1])
            assert _eq(fr.get(k), ref[key]), (k, key)
    finally:
        fr.done()
        ctx.set_tuning(0, 0)


@pytest.mark.parametrize("pipeline", [1, 2])
def test_multi_sample_frames_in_flight_have_their_own_accumulators(ctx, pipeline):
    """spp > 1 with frames in flight on alternating streams: one colour-sum buffer per frame in flight."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(512)
    w, h = 480, 270
    ctx.set_pipeline(pipeline)
    ctx.pool_upload(pool)
    ctx.resize(w, h)
    ctx.set_camera(CAMERAS["K1"])
    ctx.set_tuning(4, 9)
    fr = Frames(ctx, w, h, 8, 4)
    try:
        for k in range(8):
            fr.enqueue(k, 2 + k, 0, spp=3)
        fr.done()
        ctx.set_tuning(0, 0)
        for k in range(8):
            alone = ctx.render(None, None, None, None, 2 + k, 0, spp=3)
            assert _eq(fr.get(k), alone), k
    finally:
        fr.done()
        ctx.set_tuning(0, 0)


@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_pool_edit_between_frames_in_flight_is_not_torn(ctx, pipeline):
    """Frames A are in flight on three streams when the pool is edited (ranged svo_pool_update, then one that grows
    and re-allocates the pool); frames B follow.  A == renders of the old pool, B == renders of the new one."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(1024)
    w, h = 800, 450
    ctx.set_pipeline(pipeline)
    base = ctx.render(pool, w, h, CAMERAS["K1"], 2, 2)
    ptrs = np.unique(base["hits"]["pointer"][base["hits"]["pointer"] != 0])
    edited = pool.copy()
    edited[ptrs[::2]] = 3                                    # recolour half of the visible voxels
    grown = np.concatenate([edited, np.zeros(1 << 20, dtype=np.uint8)])   # + a tail: forces a re-allocation
    lo, hi = int(ptrs.min()), int(ptrs.max()) + 1
    ctx.set_tuning(6, 9)
    fr = Frames(ctx, w, h, 9, 3)
    try:
        for k in range(3):
            fr.enqueue(k, 2 + k, 2)
        ctx.pool_update(edited, lo, hi)                      # waits for the three frames, then edits
        for k in range(3, 6):
            fr.enqueue(k, 2 + k - 3, 2)
        ctx.pool_update(grown, pool.size, grown.size)        # grows the pool under three more frames in flight
        for k in range(6, 9):
            fr.enqueue(k, 2 + k - 6, 2)
        fr.done()
        ctx.set_tuning(0, 0)
        old = [ctx.render(pool, w, h, CAMERAS["K1"], 2 + k, 2) for k in range(3)]
        new = [ctx.render(edited, w, h, CAMERAS["K1"], 2 + k, 2) for k in range(3)]
        assert not _eq(old[0], new[0])
        for k in range(3):
            assert _eq(fr.get(k), old[k]), ("before the edit", k)
            assert _eq(fr.get(3 + k), new[k]), ("after the edit", k)
            assert _eq(fr.get(6 + k), new[k]), ("after the growing edit", k)
    finally:
        fr.done()
        ctx.set_tuning(0, 0)


@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_packed_stripes_with_samples_beyond_the_frame_height(ctx, pipeline):
    """H = 100, 3 ranks: rank 2's packed stripes occupy rows 80..111 of the gather buffer -- past H.  With spp > 1
    the colour sums are indexed like the outputs, so they must be sized for the gather buffer, not for W x H."""
    import torch
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    from svo_raytracer_amd.tiles import stripe_layout, deinterleave
    pool, _ = scene.build_scene(256)
    w, h, world = 200, 100, 3
    ctx.set_pipeline(pipeline)
    full = ctx.render(pool, w, h, CAMERAS["K1"], 2, 0, spp=3)
    rpr = stripe_layout(h, world, 0)[4]
    assert rpr * world > h
    col = torch.zeros((rpr * world, w), dtype=torch.int32, device="cuda")
    dep = torch.zeros((rpr * world, w), dtype=torch.float32, device="cuda")
    ctx.bind_outputs(col.data_ptr(), dep.data_ptr(), None)
    ctx.set_params(2, 0, 0, 0, 2, 0, 3)
    try:
        for r in range(world):
            first, step, n, out0, rows = stripe_layout(h, world, r)
            ctx.set_stripes(first, step, n, out0)
            ctx.dispatch()
        torch.cuda.synchronize()
        got = deinterleave(col.cpu().numpy().view(np.uint8).reshape(rpr * world, w, 4), world, rpr, h)
        gotd = deinterleave(dep.cpu().numpy(), world, rpr, h)
        assert (got == full["rgba"]).all()
        assert (gotd.view(np.uint32) == full["depth"].view(np.uint32)).all()
        # the same stripes into the library's own W x H images do not fit: rejected, not written out of bounds
        ctx.bind_outputs(None, None, None)
        first, step, n, out0, rows = stripe_layout(h, world, 2)
        ctx.set_stripes(first, step, n, out0)
        with pytest.raises(hiplib.SvoError) as e:
            ctx.dispatch()
        assert e.value.code == -1
        ctx.set_stripes(0, 1, 13, 0)      # the whole frame as stripes fits (last tile row partial)
        ctx.dispatch()
        assert (ctx.read_color() == full["rgba"]).all()
    finally:
        ctx.bind_outputs(None, None, None)
        ctx.set_rows(0, h)


@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_batched_dispatch_equals_one_dispatch_per_frame(ctx, pipeline):
    """svo_set_batch: one dispatch renders frameNumber .. frameNumber + n - 1 (on the persistent pipeline as ONE launch whose
    waves run from frame to frame); every frame's bytes are those of a dispatch of its own -- also with the beam
    pre-pass, with packed stripes, and with several batches in flight on different streams."""
    import torch
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    from svo_raytracer_amd.tiles import stripe_layout
    pool, _ = scene.build_scene(1024)
    w, h, nb = 640, 360, 3
    ctx.set_pipeline(pipeline)
    ctx.set_tuning(5, 9)
    ref = {}

    def alone(frame, mode, beam):
        key = (frame, mode, beam)
        if key not in ref:
            ctx.set_batch(1, 0)
            ctx.bind_outputs(None, None, None)
            ctx.set_rows(0, h)
            ref[key] = ctx.render(pool if not ref else None, w, h, CAMERAS["K1"], frame, mode, use_beam=beam)
        return ref[key]

    try:
        for mode, beam in ((0, 0), (2, 0), (0, 1)):
            for f in range(2, 2 + 2 * nb):
                alone(f, mode, beam)
            # two batches of three frames in flight on two streams
            stride = w * h
            col = [torch.zeros((nb, h, w), dtype=torch.int32, device="cuda") for _ in range(2)]
            dep = [torch.zeros((nb, h, w), dtype=torch.float32, device="cuda") for _ in range(2)]
            hit = [torch.zeros((nb, h, w, 4), dtype=torch.int32, device="cuda") for _ in range(2)]
            streams = [torch.cuda.Stream() for _ in range(2)]
            torch.cuda.synchronize()
            ctx.set_batch(nb, stride)
            for b in range(2):
                ctx.set_stream(streams[b].cuda_stream)
                ctx.bind_outputs(col[b].data_ptr(), dep[b].data_ptr(), hit[b].data_ptr())
                ctx.set_params(2 + b * nb, mode, 0, beam, 2, 0, 1)
                ctx.dispatch_async()
            torch.cuda.synchronize()
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            for b in range(2):
                for k in range(nb):
                    want = alone(2 + b * nb + k, mode, beam)
                    tag = (pipeline, mode, beam, b, k)
                    assert np.array_equal(col[b][k].cpu().numpy().view(np.uint8).reshape(h, w, 4), want["rgba"]), tag
                    assert np.array_equal(dep[b][k].cpu().numpy().view(np.uint32), want["depth"].view(np.uint32)), tag
                    got_h = hit[b][k].cpu().numpy().reshape(-1, 4).copy().view(hiplib.HIT_DTYPE).reshape(h, w)
                    assert got_h.tobytes() == want["hits"].tobytes(), tag
        # several samples per pixel in a batch: sample s of all frames is one launch; sums stay in sample order
        ctx.set_batch(1, 0)
        ctx.bind_outputs(None, None, None)
        ctx.set_rows(0, h)
        want_s = [ctx.render(None, w, h, CAMERAS["K1"], 2 + k, 0, spp=3) for k in range(nb)]
        colM = torch.zeros((nb, h, w), dtype=torch.int32, device="cuda")
        depM = torch.zeros((nb, h, w), dtype=torch.float32, device="cuda")
        ctx.bind_outputs(colM.data_ptr(), depM.data_ptr(), None)
        ctx.set_batch(nb, w * h)
        ctx.set_params(2, 0, 0, 0, 2, 0, 3)
        ctx.dispatch()
        for k in range(nb):
            assert np.array_equal(colM[k].cpu().numpy().view(np.uint8).reshape(h, w, 4), want_s[k]["rgba"]), ("spp", k)
            assert np.array_equal(depM[k].cpu().numpy().view(np.uint32), want_s[k]["depth"].view(np.uint32)), ("spp", k)
        # a batch of packed stripes (what one rank of three renders)
        first, step, n, out0, rows = stripe_layout(h, 3, 1)
        colS = torch.zeros((nb, rows, w), dtype=torch.int32, device="cuda")
        depS = torch.zeros((nb, rows, w), dtype=torch.float32, device="cuda")
        ctx.bind_outputs(colS.data_ptr(), depS.data_ptr(), None)
        ctx.set_stripes(first, step, n, 0)
        ctx.set_batch(nb, rows * w)
        ctx.set_params(2, 0, 0, 0, 2, 0, 1)
        ctx.dispatch()
        for k in range(nb):
            want = alone(2 + k, 0, 0)
            for j in range(n):
                y = (first + j * step) * 8
                ys = min(8, h - y)
                assert np.array_equal(colS[k, 8 * j:8 * j + ys].cpu().numpy().view(np.uint8).reshape(ys, w, 4), want["rgba"][y:y + ys]), (k, j)
        # a frame smaller than the 8 XCD bands (3 tile rows: five bands are empty), 5 frames per launch
        tw, th = 44, 20
        ctx.set_batch(1, 0)
        ctx.bind_outputs(None, None, None)
        ctx.set_rows(0, h)
        small = [ctx.render(None, tw, th, CAMERAS["K1"], 2 + k, 0) for k in range(5)]
        colT = torch.zeros((5, th, tw), dtype=torch.int32, device="cuda")
        depT = torch.zeros((5, th, tw), dtype=torch.float32, device="cuda")
        ctx.bind_outputs(colT.data_ptr(), depT.data_ptr(), None)
        ctx.set_batch(5, tw * th)
        ctx.set_params(2, 0, 0, 0, 2, 0, 1)
        ctx.dispatch()
        for k in range(5):
            assert np.array_equal(colT[k].cpu().numpy().view(np.uint8).reshape(th, tw, 4), small[k]["rgba"]), k
            assert np.array_equal(depT[k].cpu().numpy().view(np.uint32), small[k]["depth"].view(np.uint32)), k
        ctx.set_batch(1, 0)
        ctx.bind_outputs(None, None, None)
        ctx.resize(w, h)
        # cross-frame accumulation and batches exclude each other
        ctx.set_batch(1, 0)
        ctx.bind_outputs(colT.data_ptr(), depT.data_ptr(), None)
        ctx.resize(tw, th)
        ctx.set_progressive(True)
        ctx.set_batch(2, tw * th)
        with pytest.raises(hiplib.SvoError):
            ctx.dispatch()
        ctx.set_progressive(False)
        ctx.set_batch(1, 0)
        ctx.bind_outputs(None, None, None)
        ctx.resize(w, h)
        # library-owned images cannot hold a batch
        ctx.bind_outputs(None, None, None)
        ctx.set_rows(0, h)
        ctx.set_batch(2, w * h)
        with pytest.raises(hiplib.SvoError):
            ctx.dispatch()
    finally:
        ctx.set_progressive(False)
        ctx.set_batch(1, 0)
        ctx.bind_outputs(None, None, None)
        ctx.set_tuning(0, 0)


_ONE_LAUNCH_PER_SAMPLE = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd import hiplib
from svo_raytracer_amd.cameras import CAMERAS
pool, _ = scene.build_scene(128)
ctx = hiplib.HipContext(0, lib_path=hiplib.VARIANTS_LIB_PATH)     # (SVO_FOLD_BYTES is one of the variants library's switches)
ctx.set_pipeline(1)
out = {}
for i, (w, h, spp, bounces, mirror, frame) in enumerate([(200, 120, 5, 3, 0, 4), (64, 37, 16, 2, 0b110, 9), (333, 50, 2, 2, 0, 2)]):
    r = ctx.render(pool if i == 0 else None, w, h, CAMERAS["K1"], frame, 0, bounces=bounces, mirror_mask=mirror, spp=spp)
    out["rgba%d" % i] = r["rgba"]; out["depth%d" % i] = r["depth"]; out["hits%d" % i] = r["hits"].view(np.uint8)
np.savez(sys.argv[2], **out)
"""


def test_all_samples_in_one_launch_equal_one_launch_per_sample(ctx, tmp_path):
    """spp > 1 on the persistent pipeline: one launch carries every sample of the frame (each in a slot of its own, added
    in sample order afterwards); SVO_FOLD_BYTES=0 (a child process) forces the older one-launch-per-sample path with its
    running sums.  Same bytes, and the oracle's."""
    import os
    import subprocess
    import sys
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "per_sample.npz")
    env = dict(os.environ, SVO_FOLD_BYTES="0")
    subprocess.run([sys.executable, "-c", _ONE_LAUNCH_PER_SAMPLE, root, out], check=True, env=env, timeout=600)
    z = np.load(out)
    pool, _ = scene.build_scene(128)
    ctx.set_pipeline(1)
    for i, (w, h, spp, bounces, mirror, frame) in enumerate([(200, 120, 5, 3, 0, 4), (64, 37, 16, 2, 0b110, 9), (333, 50, 2, 2, 0, 2)]):
        got = ctx.render(pool if i == 0 else None, w, h, CAMERAS["K1"], frame, 0, bounces=bounces, mirror_mask=mirror, spp=spp)
        assert np.array_equal(got["rgba"], z["rgba%d" % i]), i
        assert np.array_equal(got["depth"].view(np.uint32), z["depth%d" % i].view(np.uint32)), i
        assert got["hits"].view(np.uint8).tobytes() == z["hits%d" % i].tobytes(), i
        ref = oracle.render(pool, w, h, CAMERAS["K1"], frame, 0, bounces=bounces, mirror_mask=mirror, spp=spp)
        assert np.array_equal(got["rgba"], ref["rgba"]), i
        assert np.array_equal(got["depth"].view(np.uint32), ref["depth"].view(np.uint32)), i
