"""Randomised differential stress of the library's ordering logic: bursts of dispatches in flight on several streams --
random pipeline, frames per dispatch, samples per pixel, beam flag, render mode, camera, tuning -- with pool edits (in place, growing)
and image-size changes between bursts; afterwards every frame of every burst must be the bytes the same state renders
one frame at a time, synchronously.  Counter sets, sample buffers, beam images and the liveness table are all re-used
round-robin behind events: an ordering hole shows up here as a torn or stale frame."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _images(col, dep, k, h, w):
    return col[k].cpu().numpy().view(np.uint8).reshape(h, w, 4), dep[k].cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("seed", [int(s) for s in __import__("os").environ.get("SVO_STRESS_SEEDS", "1,2,3,4,5,6,7,8").split(",")])
def test_random_bursts_equal_synchronous_renders(seed):
    import torch
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    rng = np.random.RandomState(seed)
    pool, _ = scene.build_scene(512)
    # even seeds: the library that ships (pipelines 0 and 1); odd seeds: libsvohip_variants.so, where pipeline 2 joins the mix
    other = [0, 2] if seed % 2 else [0, 0]
    ctx = hiplib.HipContext(0, lib_path=hiplib.VARIANTS_LIB_PATH if seed % 2 else None)
    try:
        ctx.set_pipeline(1)
        sizes = [(320, 200), (203, 131)]
        w, h = sizes[0]
        ctx.pool_upload(pool)
        ctx.resize(w, h)
        cur = pool.copy()
        streams = [torch.cuda.Stream() for _ in range(6)]
        cams = [CAMERAS["K0"], CAMERAS["K1"], CAMERAS["K2"]]
        frame = 2
        for burst in range(10):
            # ---- between bursts: sometimes edit the pool or change the image size
            r = rng.rand()
            if r < 0.3:
                ptrs = rng.randint(7, cur.size - 8, size=400)
                cur = cur.copy()
                cur[ptrs] = rng.randint(0, 5, size=400).astype(np.uint8)      # arbitrary bytes: still a legal upload
                ctx.pool_update(cur, int(ptrs.min()), int(ptrs.max()) + 1)
            elif r < 0.4:
                cur = np.concatenate([cur, np.zeros(int(rng.randint(1, 1 << 18)), dtype=np.uint8)])
                ctx.pool_update(cur, cur.size - 16, cur.size)                  # grows: re-allocation under the hood
            elif r < 0.6:
                w, h = sizes[int(rng.randint(0, 2))]
                ctx.resize(w, h)
            ctx.set_tuning(int(rng.choice([0, 3, 6, 10])), 9)
            ctx.set_pipeline(int(rng.choice([1, 1, 1] + other)))   # the burst's pipeline; the check below always uses pipeline 1
            # ---- a burst of dispatches in flight
            nd = int(rng.randint(2, 7))
            plan, bufs = [], []
            for d in range(nd):
                nb = int(rng.randint(1, 5))
                spp = int(rng.choice([1, 1, 2, 3]))
                beam = int(rng.rand() < 0.4)
                mode = int(rng.choice([0, 0, 2, 3]))
                cam = cams[int(rng.randint(0, 3))]
                col = torch.zeros((nb, h, w), dtype=torch.int32, device="cuda")
                dep = torch.zeros((nb, h, w), dtype=torch.float32, device="cuda")
                bufs.append((col, dep))
                plan.append((nb, spp, beam, mode, cam, frame))
                frame += nb
            torch.cuda.synchronize()
            for d, (nb, spp, beam, mode, cam, f0) in enumerate(plan):
                ctx.set_stream(streams[d % len(streams)].cuda_stream)
                ctx.bind_outputs(bufs[d][0].data_ptr(), bufs[d][1].data_ptr(), None)
                ctx.set_camera(cam)
                ctx.set_batch(nb, w * h)
                ctx.set_params(f0, mode, 0, beam, 2, 0, spp)
                ctx.dispatch_async()
            torch.cuda.synchronize()
            # ---- the same frames one at a time, synchronously, into the library's own images
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            ctx.bind_outputs(None, None, None)
            ctx.set_batch(1, 0)
            ctx.set_tuning(0, 0)
            ctx.set_pipeline(1)
            for d, (nb, spp, beam, mode, cam, f0) in enumerate(plan):
                for k in range(nb):
                    want = ctx.render(None, None, None, cam, f0 + k, mode, spp=spp, use_beam=beam)
                    rgba, depth = _images(bufs[d][0], bufs[d][1], k, h, w)
                    tag = (seed, burst, d, k, nb, spp, beam, mode)
                    assert np.array_equal(rgba, want["rgba"]), tag
                    assert np.array_equal(depth, want["depth"].view(np.uint32)), tag
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [int(s) for s in __import__("os").environ.get("SVO_STRESS_SEEDS", "11,12,13,14,15,16").split(",")])
def test_random_ring_submissions_with_cameras_and_sequences(seed):
    """The round-4 submission kinds on the library's own ring, mixed at random with frames in flight: batches of a static
    camera, batches whose frames carry their own cameras (per-slot camera tables fed from pinned staging, re-used behind
    events), progressive sequences (slots of per-frame colours, re-used round-robin), pool edits and ring re-creation between
    bursts.  Every frame / image must be what the same state renders synchronously, one frame at a time."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS, orbit_path
    rng = np.random.RandomState(seed)
    pool, _ = scene.build_scene(256)
    other = [0, 2] if seed % 2 else [0, 0]     # (as above: odd seeds run on the variants library, with pipeline 2 in the mix)
    ctx = hiplib.HipContext(0, lib_path=hiplib.VARIANTS_LIB_PATH if seed % 2 else None)
    ref = hiplib.HipContext(0)       # the synchronous renderer (always the library that ships): its own pool copy, table and images
    try:
        w, h = 208, 136
        cur = pool.copy()
        for c in (ctx, ref):
            c.set_pipeline(1)
            c.pool_upload(cur)
            c.resize(w, h)
        path, _ = orbit_path(64, yaw_step=0.05, pitch_step=0.01, forward=0.004)
        frame = 2
        for burst in range(8):
            if rng.rand() < 0.35:
                ptrs = rng.randint(7, cur.size - 8, size=300)
                cur = cur.copy()
                cur[ptrs] = rng.randint(0, 5, size=300).astype(np.uint8)
                for c in (ctx, ref):
                    c.pool_update(cur, int(ptrs.min()), int(ptrs.max()) + 1)
            slots, per = int(rng.randint(1, 5)), int(rng.randint(1, 6))
            ctx.set_tuning(int(rng.choice([0, 4, 10])), 9)
            ctx.set_pipeline(int(rng.choice([1, 1, 1] + other)))
            ctx.ring_create(slots, per, want_hits=False)
            plan = []
            for d in range(slots + int(rng.randint(0, slots + 1))):      # some slots are re-used within the burst
                kind = rng.choice(["static", "cams", "cams", "seq"])
                mode = int(rng.choice([0, 0, 2]))
                cam = [CAMERAS["K1"], CAMERAS["K2"]][int(rng.randint(0, 2))]
                ctx.set_camera(cam)
                if kind == "seq":
                    n = int(rng.randint(2, 7))
                    ctx.set_progressive(True)
                    ctx.set_sequence(n, fresh=True)
                    ctx.set_params(frame, mode, 0, 0, 2, 0, 1)
                    s = ctx.ring_submit(frame, 1)
                    plan.append((s, "seq", mode, cam, frame, n))
                    ctx.set_progressive(False)
                    ctx.set_sequence(1, fresh=False)
                    frame += n
                elif kind == "cams":
                    n = int(rng.randint(1, per + 1))
                    idx = rng.randint(0, 64, size=n)
                    fns = rng.choice([1, 1, 2, 5, 77], size=n).astype(np.int32)
                    ctx.set_params(3, mode, 0, 0, 2, 0, 1)
                    s = ctx.ring_submit_cams(path[idx], fns)
                    plan.append((s, "cams", mode, path[idx].copy(), fns.copy(), n))
                else:
                    n = int(rng.randint(1, per + 1))
                    ctx.set_params(frame, mode, 0, 0, 2, 0, 1)
                    s = ctx.ring_submit(frame, n)
                    plan.append((s, "static", mode, cam, frame, n))
                    frame += n
            # only the LAST submission into each slot is still there
            last = {}
            for p in plan:
                last[p[0]] = p
            for s, kind, mode, cam, f0, n in last.values():
                ctx.ring_wait(s)
                for k in range(1 if kind == "seq" else n):
                    got = ctx.ring_read(s, k)
                    tag = (seed, burst, s, kind, k, n, mode)
                    if kind == "seq":
                        ref.set_camera(cam)
                        ref.set_progressive(True)
                        ref.set_sequence(1, fresh=True)
                        for f in range(f0, f0 + n):            # one dispatch per frame, the reference's loop
                            want = ref.render(None, None, None, None, f, mode)
                            ref.set_sequence(1, fresh=False)
                        ref.set_progressive(False)
                    elif kind == "cams":
                        want = ref.render(None, None, None, cam[k], int(f0[k]), mode)
                    else:
                        want = ref.render(None, None, None, cam, f0 + k, mode)
                    assert np.array_equal(got["rgba"], want["rgba"]), tag
                    assert np.array_equal(got["depth"].view(np.uint32), want["depth"].view(np.uint32)), tag
            ctx.ring_destroy()
    finally:
        ctx.close()
        ref.close()
