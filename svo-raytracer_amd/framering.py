"""Frame ring: the per-frame driving logic of bench.py, kept importable so that the N > 1 path
(pool broadcast -> interleaved tile-row stripes -> gather to the frame owner -> de-interleave) runs
unchanged under gloo on CPU tensors in tests/ and under RCCL on the GPUs.

One step = one frame of the reference's loop (Main.updateEarly, Main.java:257-289): frameNumber is
pre-incremented (first frame = 2), the uniforms are set, the frame is dispatched.  Here `nbuf` frames
are in flight: frame k renders into buffer k % nbuf on stream k % nbuf, and -- with more than one
rank -- its bands travel to rank 0 on a separate communication stream while frame k + 1 is traced.

Gather buffer of one dispatch in flight, on every rank:  [world][planes][batch][rows_per_rank][W] 32-bit words,
planes = colour (rgba8), depth (f32 bits) [+ 4 words of hit record]; `batch` = frames per dispatch (svo_set_batch;
1 = the reference's one dispatch per frame).  A rank renders its stripes PACKED into its own chunk (svo_set_stripes
with out_row0 = 0, outputs bound at the chunk, frame k of the batch rows_per_rank * W elements further), so one
contiguous chunk per rank and ONE gather per dispatch carry colour and depth of all its frames together.

Streams, launch events and the submission itself belong to the library's own ring (include/svo_hip.h, svo_ring_*:
one slot per dispatch in flight, a HIP stream of its own per slot); what stays here is what torch is needed for --
the gather buffers RCCL reads in place (bound to the slots with svo_ring_bind_slot), the gather on a communication
stream ordered against the slots' streams, and the de-interleave on the frame owner.  With one rank and no what-if
layout nothing is bound: the slots render into the library's own images and frames are read back through
svo_ring_read_*.

The renderer is duck-typed (hiplib.HipContext on the GPU; tests pass a CPU stand-in):
  set_stripes(...) set_params(...) ring_create(slots, frames, want_hits) ring_bind_slot(slot, c, d, h, stride)
  ring_submit(frame, n) -> slot  ring_wait(slot)  ring_query(slot)  ring_device_ptrs(slot)  ring_read(slot, k, want_hits)
"""
import torch

from .tiles import TILE, stripe_layout


class _NoStream:
    """CPU stand-in for a torch.cuda stream / event (everything is synchronous there)."""
    cuda_stream = 0

    def wait_event(self, ev):
        pass

    def wait_stream(self, s):
        pass

    def record(self, s=None):
        pass


class FrameRing:
    def __init__(self, renderer, width, height, world=1, rank=0, nbuf=3, device="cuda", dist=None,
                 want_hits=False, force_comm=False, first_frame=2, params=None, as_rank=None, batch=1, exchange="rccl",
                 advance=True):
        """params: dict(render_mode, buffer_end, use_beam, bounces, mirror_mask, spp) -- constant over the run.
        as_rank = (r, n): render what rank r of n would, without any communication (single-GPU what-if runs).
        exchange: how a rank's chunk reaches the frame owner -- "rccl": one gather per dispatch on a communication stream
        (send / receive kernels, which need CU slots next to the persistent waves); "copy": the library forwards every
        slot into the owner's gather buffer (opened through its IPC handle) with a device-to-device copy behind the launch,
        and a sequence word per rank tells the owner when it has landed (svo_ring_forward_slot; SDMA between GPUs)."""
        self.r = renderer
        self.W, self.H = int(width), int(height)
        self.world, self.rank = int(world), int(rank)
        self.nbuf = int(nbuf)
        self.dist = dist
        self.cuda = torch.device(device).type == "cuda"
        self.device = device
        self.exchange = exchange if (dist is not None and self.world > 1) else "rccl"
        self.copy_mode = self.exchange == "copy"
        self.use_comm = dist is not None and (self.world > 1 or force_comm) and not self.copy_mode
        self.force_comm = force_comm
        self.planes = 2 + (4 if want_hits else 0)
        self.want_hits = want_hits
        self.params = dict(render_mode=0, buffer_end=0, use_beam=0, bounces=2, mirror_mask=0, spp=1)
        self.params.update(params or {})
        self.k = 0
        self.batch = max(1, int(batch))
        self.first_frame = int(first_frame)
        self.advance = bool(advance)   # False: every submission starts at first_frame again (a progressive sequence per step)
        self.path = None               # (cams [n][15], frame numbers [n]): frames with their own cameras (start_path)
        lw, lr = (self.world, self.rank) if as_rank is None else (int(as_rank[1]), int(as_rank[0]))
        self.layout_world = lw
        self.s_first, self.s_step, self.s_n, _, self.rows_per_rank = stripe_layout(self.H, lw, lr)
        self.chunk_world = self.world if as_rank is None else 1
        rpr = self.rows_per_rank
        # the library's ring: one slot per dispatch in flight
        if hasattr(self.r, "set_stripes"):
            self.r.set_stripes(self.s_first, self.s_step, self.s_n, 0)
        p = self.params
        self.r.set_params(self.first_frame, p["render_mode"], p["buffer_end"], p["use_beam"], p["bounces"], p["mirror_mask"], p["spp"])
        self.r.ring_create(self.nbuf, self.batch, want_hits)
        # torch-owned gather buffers, [world][planes][batch][rpr][W] words; rank r's chunk is self.buf[b][r].  Not needed
        # (library-owned images instead) when this process renders whole frames for itself.
        self.own_images = (not self.use_comm) and as_rank is None and self.world == 1
        self.gbuf, self.remote = None, None
        if self.copy_mode:
            self._setup_copy_exchange(rpr)
        self.buf = None if (self.own_images or (self.copy_mode and self.rank == 0)) else [
            torch.zeros((self.chunk_world, self.planes, self.batch, rpr, self.W), dtype=torch.int32, device=device)
            for _ in range(self.nbuf)]
        self.cams_of = [None] * self.nbuf           # per slot: the cameras of its frames when they carry their own (start_path)
        self.frame_of = [None] * self.nbuf          # first frameNumber held by each slot (a list of them on a camera path)
        self.count_of = [0] * self.nbuf             # frames it holds
        self.scratch = (torch.zeros((self.planes, self.batch, rpr, self.W), dtype=torch.int32, device=device)
                        if self.use_comm and self.rank == 0 else None)
        self.my_chunk = self.rank if as_rank is None else 0
        if self.cuda:
            torch.cuda.synchronize()                # the buffers are zeroed before a slot's stream writes them
        if self.copy_mode and self.rank != 0:
            self.chunk_world, self.my_chunk = 1, 0       # only this rank's chunk lives here; it travels by copy
            self.buf = [torch.zeros((1, self.planes, self.batch, rpr, self.W), dtype=torch.int32, device=device)
                        for _ in range(self.nbuf)]
            if self.cuda:
                torch.cuda.synchronize()
        if self.buf is not None or self.copy_mode:
            for b in range(self.nbuf):
                c, d, h = self._ptrs(b)
                self.r.ring_bind_slot(b, c, d, h, rpr * self.W)
                if self.copy_mode and self.rank != 0:
                    cb = self.chunk_bytes
                    self.r.ring_forward_slot(b, c, self.remote[b] + self.rank * cb, cb, self.remote[b] + self.world * cb + 4 * self.rank)
        if self.cuda and self.use_comm:
            # the slots' streams, wrapped so that torch events can order the gather against them.  HIP maps streams onto
            # a small number of hardware queues (GPU_MAX_HW_QUEUES, 4 by default): two frame streams on one queue
            # serialise their launches (3.6 instead of 4.5 Grays/s, tools/history/r02_streams.sh) -- bench.py sets 8
            # before it touches the GPU, svo_create does when it comes first.
            self.streams = [torch.cuda.ExternalStream(self.r.ring_device_ptrs(b)["stream"]) for b in range(self.nbuf)]
            self.comm_stream = torch.cuda.Stream()
        else:
            self.streams = [_NoStream() for _ in range(self.nbuf)]
            self.comm_stream = _NoStream() if self.use_comm else None
        self.gathered = [None] * self.nbuf          # event: the gather that last read buffer b has finished
        self.last_seq = [0] * self.nbuf             # number (1, 2, ...) of the last dispatch into buffer b
        self.timing = False
        self.launch_ms = []                         # (GPU ms, frames) of every timed submission (svo_ring_query)
        self.timed_of = [False] * self.nbuf
        self.gather_events = []                     # (start, end) events of every timed gather, on the communication stream
        import os
        self.host_wait = os.environ.get("SVO_RING_HOST_WAIT", "1") != "0"   # experiment knob
        self.dispatches = 0

    def _setup_copy_exchange(self, rpr):
        """rank 0 owns one gather buffer per slot, [world chunks][world sequence words], allocated by the library (a
        whole allocation, so that its IPC handle maps it from its first byte) and opened by every other rank"""
        self.chunk_bytes = self.planes * self.batch * rpr * self.W * 4
        total = self.world * self.chunk_bytes + 4 * self.world
        handles = [None] * self.nbuf
        if self.rank == 0:
            self.gbuf = [self.r.dev_alloc(total) for _ in range(self.nbuf)]
            handles = [self.r.ipc_export(p) for p in self.gbuf]
        self.dist.broadcast_object_list(handles, src=0)
        if self.rank != 0:
            self.remote = [self.r.ipc_open(h) for h in handles]

    def landed(self, b, timeout_s=20.0):
        """frame owner, copy exchange: wait until every rank's chunk of the last dispatch into buffer b has arrived"""
        import time
        import numpy as np
        if not (self.copy_mode and self.rank == 0) or self.last_seq[b] == 0:
            return True
        t0 = time.time()
        while True:
            flags = self.r.dev_read(self.gbuf[b] + self.world * self.chunk_bytes, 4 * self.world, dtype=np.uint32)
            if any(int(flags[r]) > self.last_seq[b] for r in range(1, self.world)):
                # the word carries the SENDER's submission count: a rank that submitted into this slot again before the owner
                # read it has overwritten the frames waited for (include/svo_hip.h, svo_ring_forward_slot: LOCKSTEP)
                raise RuntimeError("copy exchange: buffer %d holds a later submission than the owner waits for (sequence %d, "
                                   "flags %s): ranks must submit in lockstep" % (b, self.last_seq[b], flags.tolist()))
            if all(int(flags[r]) == self.last_seq[b] for r in range(1, self.world)):
                return True
            if time.time() - t0 > timeout_s:
                raise RuntimeError("copy exchange: buffer %d still waits for ranks %s (sequence %d, flags %s)" % (
                    b, [r for r in range(1, self.world) if int(flags[r]) < self.last_seq[b]], self.last_seq[b], flags.tolist()))
            time.sleep(0.0005)

    # ---- one frame ------------------------------------------------------------------------------
    def _ptrs(self, b):
        word = 4
        if self.copy_mode and self.rank == 0:
            base = self.gbuf[b]                       # the owner renders its chunk (chunk 0) in place
        else:
            base = self.buf[b][self.my_chunk].data_ptr()
        plane = self.batch * self.rows_per_rank * self.W * word
        return base, base + plane, (base + 2 * plane if self.want_hits else None)

    def start_path(self, cams, frame_numbers):
        """From now on every frame carries its own camera and frameNumber (svo_ring_submit_cams): frame i of the path is
        cams[i % len], frame_numbers[i % len].  None: back to the context's camera."""
        self.path = None if cams is None else (cams, frame_numbers)
        self.path_k = 0

    def step(self, nframes=None):
        """Submit the next `nframes` frames (default: a whole batch) as one dispatch; returns the first frameNumber."""
        n = self.batch if nframes is None else max(1, min(int(nframes), self.batch))
        b = self.dispatches % self.nbuf
        frame = self.first_frame + (self.k if self.advance else 0)
        # a slot is re-used once its previous frames are complete (a host that reads them has waited for them anyway);
        # the other nbuf - 1 submissions keep the GPU busy meanwhile
        if self.dispatches >= self.nbuf and self.host_wait:
            self.r.ring_wait(b)
            if self.timing and self.timed_of[b]:
                self.launch_ms.append((self.r.ring_query(b)["gpu_ms"], self.count_of[b]))
        self.dispatches += 1
        self.k += n
        if self.gathered[b] is not None:
            self.streams[b].wait_event(self.gathered[b])
        if self.path is not None:
            cams, fns = self.path
            idx = [(self.path_k + i) % len(fns) for i in range(n)]
            self.path_k += n
            self.cams_of[b] = [cams[i] for i in idx]
            frame = [int(fns[i]) for i in idx]
            slot = self.r.ring_submit_cams([cams[i] for i in idx], frame)
        else:
            self.cams_of[b] = None
            slot = self.r.ring_submit(frame, n)
        assert slot == b, (slot, b)
        self.last_seq[b] = self.dispatches
        self.frame_of[b] = frame
        self.count_of[b] = n
        self.timed_of[b] = self.timing
        if self.use_comm:
            self._gather(b, self.streams[b])
        return frame

    def _gather(self, b, stream):
        dist = self.dist
        if self.cuda:
            done = torch.cuda.Event()
            done.record(stream)
            ctxmgr = torch.cuda.stream(self.comm_stream)
        else:
            done = None
            import contextlib
            ctxmgr = contextlib.nullcontext()
        with ctxmgr:
            if done is not None:
                self.comm_stream.wait_event(done)
            t0 = None
            if self.cuda and self.timing:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record(self.comm_stream)
            full = self.buf[b]
            mine = full[self.rank]
            if self.rank == 0:
                parts = [full[r] for r in range(self.world)]
                parts[0] = self.scratch   # the owner's chunk is already in place; gather needs a slot for it
                dist.gather(mine, gather_list=parts, dst=0)
            else:
                dist.gather(mine, gather_list=None, dst=0)
            if self.cuda:
                ev = torch.cuda.Event(enable_timing=self.timing)
                ev.record(self.comm_stream)
                self.gathered[b] = ev
                if self.timing:
                    self.gather_events.append((t0, ev))

    # ---- after the run ----------------------------------------------------------------------------
    def drain(self):
        for b in range(min(self.nbuf, self.dispatches)):
            self.r.ring_wait(b)
            if self.timed_of[b]:
                self.launch_ms.append((self.r.ring_query(b)["gpu_ms"], self.count_of[b]))
                self.timed_of[b] = False
        if self.cuda:
            torch.cuda.synchronize()
        for b in range(self.nbuf):
            self.landed(b)

    def gather_ms(self):
        """mean GPU milliseconds of a timed gather (events on the communication stream, after drain()); None without one"""
        if not self.gather_events:
            return None
        ms = [a.elapsed_time(b) for a, b in self.gather_events]
        return round(sum(ms) / len(ms), 4)

    def frame_images(self, b, k=0):
        """(frameNumber, colour [H][W] int32, depth [H][W] float32[, hits [H][W][4] int32]) of frame k of buffer b in
        frame order, on the frame owner (rank 0) after drain(); with as_rank only the rows that rank rendered are valid."""
        if self.own_images:                     # whole frames in the library's own images: read back through the ring
            img = self.r.ring_read(b, k, want_hits=self.want_hits)
            out = (self.frame_number(b, k), torch.from_numpy(img["rgba"].view("<i4").reshape(self.H, self.W)),
                   torch.from_numpy(img["depth"]))
            if self.want_hits:
                out = out + (torch.from_numpy(img["hits"].view("<i4").reshape(self.H, self.W, 4)),)
            return out
        if self.copy_mode and self.rank == 0:   # the gather buffer is the library's: read it back, then as below
            import numpy as np
            self.landed(b)
            raw = self.r.dev_read(self.gbuf[b], self.world * self.chunk_bytes, dtype=np.int32)
            whole = torch.from_numpy(raw.reshape(self.world, self.planes, self.batch, self.rows_per_rank, self.W).copy())
        else:
            whole = self.buf[b]
        full = whole[:, :, k]                   # [chunks][planes][rpr][W]
        per = self.rows_per_rank // TILE
        cw = full.shape[0]
        lw = self.layout_world

        def order(x):                           # x: [chunks][rpr][W...] stripe-major -> frame order
            if cw == lw:
                v = x.reshape(cw, per, TILE, *x.shape[2:])
                v = v.permute(1, 0, 2, *range(3, v.dim()))
                return v.reshape(per * cw * TILE, *x.shape[2:])[: self.H]
            # a single chunk of a wider layout (as_rank): scatter its tile rows to where they belong
            out = torch.zeros((per * lw * TILE,) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
            v = out.reshape(per, lw, TILE, *x.shape[2:])
            v[:, self.s_first] = x[0].reshape(per, TILE, *x.shape[2:])
            return out[: self.H]

        color = order(full[:, 0])
        depth = order(full[:, 1]).view(torch.float32)
        if self.want_hits:
            # the hit image is pixel-major (16 bytes per pixel) inside the chunk's last four plane-sized slots
            hits = order(whole[:, 2:6].reshape(cw, self.batch, self.rows_per_rank, self.W, 4)[:, k])
            return self.frame_number(b, k), color, depth, hits
        return self.frame_number(b, k), color, depth

    def frame_number(self, b, k):
        f = self.frame_of[b]
        return f[k] if isinstance(f, list) else f + k

    def rendered_rows_mask(self):
        """bool [H]: rows this rank's stripes cover (all rows on the owner after a gather)."""
        m = torch.zeros(self.H, dtype=torch.bool)
        if self.use_comm or self.layout_world == 1:
            m[:] = True
            return m
        for j in range(self.s_n):
            y = (self.s_first + j * self.s_step) * TILE
            m[y:y + TILE] = True
        return m


def replicate_pool(dist, pool_np, rank, world, device="cuda"):
    """The pool is built once on rank 0 and replicated by one broadcast.  Returns a uint8 tensor on
    `device` holding the pool on every rank."""
    n = torch.tensor([int(pool_np.size) if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, 0)
    nbytes = int(n.item())
    dpool = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if rank == 0:
        dpool.copy_(torch.from_numpy(pool_np))
    dist.broadcast(dpool, 0)
    return dpool


class GroupAsContext:
    """hiplib.HipGroup behind the calls bench.py makes on a HipContext (bench.py --driver group): one process, n member
    GPUs inside the library (include/svo_hip.h, svo_group_*).  What is per GPU fans out to the members."""

    def __init__(self, group):
        self.g = group
        self.n = group.n

    def build_from_heightmap(self, hmap, mmap):
        return self.g.build_from_heightmap(hmap, mmap)

    def pool_download(self, nbytes):
        return self.g.pool_download(nbytes)

    def resize(self, w, h):
        self.g.resize(w, h)

    def set_camera(self, cam):
        self.g.set_camera(cam)

    def set_pipeline(self, p):
        self.g.set_pipeline(p)

    def set_tuning(self, w, t):
        self.g.set_tuning(w, t)

    def set_progressive(self, on):
        self.g.set_progressive(on)

    def set_sequence(self, n, fresh=True):
        self.g.set_sequence(n, fresh)

    def set_hit_records(self, on):
        for r in range(self.n):
            self.g.member(r).set_hit_records(on)

    def set_reserved_cus(self, n):
        pass                                   # peer copies need no CU slots

    def set_batch(self, n, stride):
        for r in range(self.n):
            self.g.member(r).set_batch(n, stride)

    def set_params(self, *a):
        self.g.set_params(*a)

    def derived_info(self):
        infos = [self.g.member(r).derived_info() for r in range(self.n)]   # builds every member's table
        return infos[0]

    def count_frame(self):
        tot = None
        for r in range(self.n):                # every member counts its own stripes; the sum is the frame's
            c = self.g.member(r).count_frame()
            tot = c if tot is None else {k: (tot[k] + c[k] if k not in ("max_iter", "last_dispatch_ms", "device") else max(tot[k], c[k]))
                                         for k in c}
        return tot

    def close(self):
        self.g.close()


class GroupRing:
    """FrameRing's interface over svo_group_ring_*: the exchange, the streams and the de-interleave are the library's."""

    def __init__(self, group, width, height, nbuf=3, want_hits=False, first_frame=2, batch=1, exchange="copy", advance=True):
        self.g, self.W, self.H = group, int(width), int(height)
        self.nbuf, self.batch, self.want_hits = int(nbuf), max(1, int(batch)), want_hits
        self.first_frame, self.advance = int(first_frame), bool(advance)
        self.world = group.n
        tile_rows = (self.H + TILE - 1) // TILE
        self.rows_per_rank = ((tile_rows + self.world - 1) // self.world) * TILE
        group.ring_create(self.nbuf, self.batch, want_hits, 1 if exchange == "rccl" else 0)
        self.k = self.dispatches = 0
        self.frame_of, self.count_of, self.cams_of = [None] * self.nbuf, [0] * self.nbuf, [None] * self.nbuf
        self.timed_of = [False] * self.nbuf
        self.timing, self.launch_ms, self.path = False, [], None

    def start_path(self, cams, frame_numbers):
        self.path = None if cams is None else (cams, frame_numbers)
        self.path_k = 0

    def step(self, nframes=None):
        n = self.batch if nframes is None else max(1, min(int(nframes), self.batch))
        b = self.dispatches % self.nbuf
        frame = self.first_frame + (self.k if self.advance else 0)
        if self.dispatches >= self.nbuf:
            self.g.ring_wait(b)
            if self.timing and self.timed_of[b]:
                self.launch_ms.append((self.g.ring_query(b)["gpu_ms"], self.count_of[b]))
        self.dispatches += 1
        self.k += n
        if self.path is not None:
            cams, fns = self.path
            idx = [(self.path_k + i) % len(fns) for i in range(n)]
            self.path_k += n
            self.cams_of[b] = [cams[i] for i in idx]
            frame = [int(fns[i]) for i in idx]
            slot = self.g.ring_submit_cams([cams[i] for i in idx], frame)
        else:
            self.cams_of[b] = None
            slot = self.g.ring_submit(frame, n)
        assert slot == b, (slot, b)
        self.frame_of[b], self.count_of[b], self.timed_of[b] = frame, n, self.timing
        return frame

    def drain(self):
        for b in range(min(self.nbuf, self.dispatches)):
            self.g.ring_wait(b)
            if self.timed_of[b]:
                self.launch_ms.append((self.g.ring_query(b)["gpu_ms"], self.count_of[b]))
                self.timed_of[b] = False

    def gather_ms(self):
        return None

    def frame_number(self, b, k):
        f = self.frame_of[b]
        return f[k] if isinstance(f, list) else f + k

    def frame_images(self, b, k=0):
        img = self.g.ring_read(b, k, want_hits=self.want_hits)
        out = (self.frame_number(b, k), torch.from_numpy(img["rgba"].view("<i4").reshape(self.H, self.W)), torch.from_numpy(img["depth"]))
        if self.want_hits:
            out = out + (torch.from_numpy(img["hits"].view("<i4").reshape(self.H, self.W, 4)),)
        return out

    def rendered_rows_mask(self):
        return torch.ones(self.H, dtype=torch.bool)
