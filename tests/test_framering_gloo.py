"""bench.py's own N > 1 logic (svo_raytracer_amd/framering.py) on CPU: world_size-2 and -3 gloo runs of
pool broadcast -> interleaved tile-row stripes -> one gather per frame to rank 0 -> de-interleave, with
frameNumber advancing every step and several frames in the ring.  The renderer here is a CPU stand-in
that fills the bound buffers from the CPU oracle (test infrastructure): what is under test is the
sharding, buffer layout, ring bookkeeping and reassembly, not the kernel."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleStripeRenderer:
    """Duck-types the part of hiplib.HipContext that FrameRing drives: the svo_ring_* calls, with caller-owned slots
    (the ring's own images only matter on the GPU)."""

    def __init__(self, pool, w, h, cam):
        self.pool, self.w, self.h, self.cam = pool, w, h, cam
        self.stripes = (0, 1, (h + 7) // 8, 0)
        self.params = None
        self.calls = []
        self.slots, self.per, self.next = [], 0, 0

    def set_stripes(self, first, step, n, out0):
        self.stripes = (first, step, n, out0)

    def set_params(self, frame, mode, buffer_end, use_beam, bounces, mirror, spp):
        self.params = (frame, mode, bounces, mirror, spp)

    def ring_create(self, slots, frames_per_slot, want_hits):
        self.slots = [None] * slots
        self.per, self.next = frames_per_slot, 0

    def ring_bind_slot(self, slot, c, d, h, stride):
        self.slots[slot] = (c, d, h, stride)

    def ring_submit(self, frame, n):
        assert 1 <= n <= self.per
        slot = self.next % len(self.slots)
        self.next += 1
        c, d, h, stride = self.slots[slot]
        self.ptrs = (c, d, h)
        for k in range(n):        # frame k of the submission: frameNumber + k, outputs `stride` elements further
            self._one(frame + k, k * stride)
        return slot

    def ring_submit_cams(self, cams, frame_numbers):
        n = len(frame_numbers)
        assert 1 <= n <= self.per
        slot = self.next % len(self.slots)
        self.next += 1
        c, d, h, stride = self.slots[slot]
        self.ptrs = (c, d, h)
        keep = self.cam
        for k in range(n):        # frame k of the submission: its own camera and frameNumber
            self.cam = cams[k]
            self._one(int(frame_numbers[k]), k * stride)
        self.cam = keep
        return slot

    def ring_wait(self, slot):
        pass

    def ring_query(self, slot):
        return {"done": True, "gpu_ms": 1.0}

    def ring_device_ptrs(self, slot):
        return {"stream": 0}

    def _one(self, frame, eoff):
        from oracle import oracle
        first, step, n, out0 = self.stripes
        _, mode, bounces, mirror, spp = self.params
        self.calls.append(frame)
        cptr, dptr, hptr = self.ptrs
        rows = out0 + 8 * n
        col = np.ctypeslib.as_array((ctypes.c_uint32 * (rows * self.w)).from_address(cptr + 4 * eoff)).reshape(rows, self.w)
        dep = np.ctypeslib.as_array((ctypes.c_float * (rows * self.w)).from_address(dptr + 4 * eoff)).reshape(rows, self.w)
        hit = None
        if hptr:
            hit = np.ctypeslib.as_array((ctypes.c_uint32 * (rows * self.w * 4)).from_address(hptr + 16 * eoff)).reshape(rows, self.w, 4)
        for j in range(n):
            y0 = (first + j * step) * 8
            y1 = min(self.h, y0 + 8)
            if y0 >= self.h:
                continue
            r = oracle.render(self.pool, self.w, self.h, self.cam, frame, mode, bounces=bounces, mirror_mask=mirror,
                              spp=spp, rows=(y0, y1), want_hits=hit is not None)
            o = out0 + 8 * j
            col[o:o + (y1 - y0)] = r["rgba"].view(np.uint32).reshape(self.h, self.w)[y0:y1]
            dep[o:o + (y1 - y0)] = r["depth"][y0:y1]
            if hit is not None:
                hit[o:o + (y1 - y0)] = r["hits"].view(np.uint32).reshape(self.h, self.w, 4)[y0:y1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, nbuf, steps, want_hits, out_path, batch=1):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from svo_raytracer_amd.framering import FrameRing, replicate_pool
    pool = scene.build_scene(64)[0] if rank == 0 else None          # built on rank 0 only ...
    dpool = replicate_pool(dist, pool, rank, world, device="cpu")   # ... and replicated by one broadcast
    rend = OracleStripeRenderer(dpool.numpy(), w, h, CAMERAS["K1"])
    ring = FrameRing(rend, w, h, world=world, rank=rank, nbuf=nbuf, device="cpu", dist=dist, want_hits=want_hits,
                     first_frame=2, params=dict(render_mode=0, buffer_end=int(dpool.numel())), batch=batch)
    left = steps
    while left > 0:                                    # `steps` frames, the last dispatch a partial batch if need be
        n = min(batch, left)
        ring.step(n)
        left -= n
    ring.drain()
    assert rend.calls == list(range(2, 2 + steps))    # frameNumber pre-incremented once per frame (Main.java:275)
    if rank == 0:
        out, i = {}, 0
        for b in range(nbuf):
            for k in range(ring.count_of[b]):
                imgs = ring.frame_images(b, k)
                out["frame%d" % i] = np.int64(imgs[0])
                out["color%d" % i] = imgs[1].numpy()
                out["depth%d" % i] = imgs[2].numpy()
                if want_hits:
                    out["hits%d" % i] = imgs[3].numpy()
                i += 1
        out["nframes"] = np.int64(i)
        np.savez(out_path, **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h,nbuf,steps,want_hits,batch", [(2, 96, 3, 5, False, 1), (2, 100, 2, 3, True, 1),
                                                                   (3, 116, 3, 4, False, 1), (2, 100, 2, 7, True, 3),
                                                                   (8, 136, 4, 12, False, 5)])   # bench.py's defaults at N = 8
def test_frame_ring_over_gloo_reassembles_every_frame_in_the_ring(tmp_path, world, h, nbuf, steps, want_hits, batch):
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    w = 64
    out = str(tmp_path / "ring.npz")
    mp.spawn(_worker, args=(world, _free_port(), w, h, nbuf, steps, want_hits, out, batch), nprocs=world, join=True)
    z = np.load(out)
    pool, _ = scene.build_scene(64)
    seen = set()
    for b in range(int(z["nframes"])):
        frame = int(z["frame%d" % b])
        seen.add(frame)
        full = oracle.render(pool, w, h, CAMERAS["K1"], frame, 0)
        assert (z["color%d" % b].view(np.uint8).reshape(h, w, 4) == full["rgba"]).all(), (b, frame)
        assert (z["depth%d" % b].view(np.uint32) == full["depth"].view(np.uint32)).all(), (b, frame)
        if want_hits:
            assert (z["hits%d" % b].astype(np.uint32).reshape(h, w, 4) == full["hits"].view(np.uint32).reshape(h, w, 4)).all()
    # the ring holds the frames of the last nbuf dispatches of the run, all different
    if batch == 1:
        assert seen == set(range(2 + steps - nbuf, 2 + steps))
    else:
        assert len(seen) == int(z["nframes"]) and max(seen) == 1 + steps


def test_what_if_rank_layout_matches_the_real_split():
    """--as-rank r/n (single-GPU what-if runs of bench.py): the rows a lone rank renders are the rows it would own."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from svo_raytracer_amd.framering import FrameRing
    from oracle import oracle
    pool, _ = scene.build_scene(64)
    w, h = 48, 100
    full = oracle.render(pool, w, h, CAMERAS["K1"], 2, 0)
    cover = np.zeros(h, dtype=np.int32)
    for r in range(3):
        rend = OracleStripeRenderer(pool, w, h, CAMERAS["K1"])
        ring = FrameRing(rend, w, h, nbuf=1, device="cpu", as_rank=(r, 3), params=dict(render_mode=0))
        ring.step()
        fr, col, dep = ring.frame_images(0)
        m = ring.rendered_rows_mask().numpy()
        cover += m
        assert (col.numpy().view(np.uint8).reshape(h, w, 4)[m] == full["rgba"][m]).all()
        assert (dep.numpy().view(np.uint32)[m] == full["depth"].view(np.uint32)[m]).all()
    assert (cover == 1).all()


def test_bench_presets_and_pmc_gating():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse([])
    assert (a.size, a.width, a.height, a.mode, a.bounces, a.spp, a.scaling) == (8192, 1920, 1080, 0, 2, 1, "strong")
    a = bench.parse(["--config", "C4", "--gpus", "8"])
    assert (a.width, a.height, a.bounces, a.mirror) == (3840, 2160, 5, 0b1000) and a.gpus == 8
    a = bench.parse(["--config", "C5", "--spp", "16"])
    assert a.spp == 16 and a.mode == 0
    a = bench.parse(["--config", "C2"])
    assert (a.size, a.mode) == (2048, 1)
    h = bench.source_hash()
    assert len(h) == 16 and h == bench.source_hash()
    assert bench.pmc_for("no such key") is None
