"""bench.py's main() itself -- argument handling, scene replication, stripe split, ray counting, the timing protocol
(barrier, MAX over ranks), verification against the oracle, the JSON line -- on CPU under gloo, world size 2, with a
stand-in for hiplib.HipContext that renders with the CPU oracle (test infrastructure; the product path has no such
fallback).  What the GPU tests cannot cover on a one-GPU box: main() with more than one rank."""
import ctypes
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from test_framering_gloo import OracleStripeRenderer   # noqa: E402


class StubContext(OracleStripeRenderer):
    """The calls bench.main() makes on a HipContext, answered by the oracle."""

    def __init__(self, local_rank):
        super().__init__(None, 0, 0, None)
        self.own = {}

    def build_from_heightmap(self, hmap, mmap):
        import svo_raytracer_amd.scene as scene
        self.pool = scene.build_scene(hmap.shape[0])[0]
        return int(self.pool.size)

    def pool_download(self, nbytes):
        return self.pool[:nbytes].copy()

    def pool_upload_device(self, ptr, nbytes):
        self.pool = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(ptr)).copy()

    def resize(self, w, h):
        self.w, self.h = w, h
        self.stripes = (0, 1, (h + 7) // 8, 0)

    def set_camera(self, cam):
        self.cam = cam

    def set_pipeline(self, p): pass
    def set_tuning(self, w, t): pass
    def set_hit_records(self, on): pass
    def set_reserved_cus(self, n): pass
    def set_batch(self, n, stride): pass
    def close(self): pass

    def count_frame(self):
        from oracle import oracle
        frame, mode, bounces, mirror, spp = self.params
        first, step, n, _ = self.stripes
        tot = dict(rays=0, iterations=0, alg_bytes=0, pixels=0, nan_rays=0)
        for j in range(n):
            y0 = (first + j * step) * 8
            if y0 >= self.h:
                continue
            st = oracle.render(self.pool, self.w, self.h, self.cam, frame, mode, bounces=bounces, mirror_mask=mirror, spp=spp,
                               rows=(y0, min(self.h, y0 + 8)), want_hits=False)["stats"]
            for k in tot:
                tot[k] += st.get(k, 0)
        return tot

    def time_frames(self, warm, iters):
        return np.full(iters, 1.0, np.float32)

    # library-owned slots (FrameRing uses them with one rank)
    def ring_create(self, slots, frames_per_slot, want_hits):
        super().ring_create(slots, frames_per_slot, want_hits)
        n = self.w * self.h
        self.own = {b: (np.zeros(frames_per_slot * n, np.uint32), np.zeros(frames_per_slot * n, np.float32)) for b in range(slots)}
        for b, (c, d) in self.own.items():
            self.slots[b] = (c.ctypes.data, d.ctypes.data, None, n)

    def ring_read(self, slot, k, want_hits=False):
        c, d = self.own[slot]
        n = self.w * self.h
        return {"rgba": c[k * n:(k + 1) * n].view(np.uint8).reshape(self.h, self.w, 4).copy(),
                "depth": d[k * n:(k + 1) * n].reshape(self.h, self.w).copy()}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, argv, out_path):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import bench
    rc, line = bench.main(argv, ctx_factory=StubContext)
    assert rc == 0
    if rank == 0:
        json.dump(line, open(out_path, "w"))
    if dist.is_initialized():
        dist.destroy_process_group()


ARGS = ["--size", "64", "--width", "72", "--height", "50", "--steps", "5", "--warmup", "2", "--inflight", "2", "--batch", "2",
        "--cpu-seconds", "0", "--long-steps", "6"]


@pytest.mark.parametrize("world,scaling", [(2, "strong"), (2, "weak"), (3, "strong")])
def test_bench_main_under_gloo(tmp_path, world, scaling):
    out = str(tmp_path / "line.json")
    argv = ARGS + ["--gpus", str(world), "--scaling", scaling]
    mp.spawn(_worker, args=(world, _free_port(), argv, out), nprocs=world, join=True)
    line = json.load(open(out))
    assert line["verified"] is True and line["n_gpus"] == world and line["ranks_seen"] == world
    assert line["steps"] == 5 and line["warmup"] == 2 and line["frames_in_flight"] == 4
    assert line["rank_ms_per_step"]["min"] <= line["rank_ms_per_step"]["max"] and abs(line["ms_per_step"] - line["rank_ms_per_step"]["max"]) < 1e-3
    # rays: every rank counted its own stripes; the sum is the whole frame's
    from oracle import oracle
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    pool = scene.build_scene(64)[0]
    h = 50 * world if scaling == "weak" else 50
    want = [oracle.render(pool, 72, h, CAMERAS["K1"], f, 0, want_hits=False)["stats"]["rays"] for f in (4, 8)]
    assert line["config"]["rays_per_frame"] == int(round(sum(want) / 2.0))
    assert line["value"] > 0 and line["unit"] == "Mrays/s" and line["scaling"] == scaling
    # the second, longer timed region of an N > 1 run, as an extra key
    assert line["value_long_run"] > 0 and line["long_run_steps"] == 6 and line["driver"].startswith("torch.distributed")


def test_bench_main_camera_path_under_gloo(tmp_path):
    """--camera-path orbit at world size 2: every frame with its own camera and frameNumber 1, verified per frame"""
    out = str(tmp_path / "line.json")
    argv = ARGS + ["--gpus", "2", "--camera-path", "orbit"]
    mp.spawn(_worker, args=(2, _free_port(), argv, out), nprocs=2, join=True)
    line = json.load(open(out))
    assert line["verified"] is True and "its own camera" in line["config"]["verification"]
    assert "moving along path" in line["config"]["workload"] and line["value"] > 0


def test_bench_main_one_rank_stub(tmp_path):
    """one rank, no launcher: library-owned slots, frames read back through the ring"""
    out = str(tmp_path / "line.json")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    import bench
    rc, line = bench.main(ARGS + ["--gpus", "1"], ctx_factory=StubContext)
    assert rc == 0 and line["verified"] is True and line["ranks_seen"] == 1 and line["gather_ms"] is None
    # (value_one_frame_at_a_time is the wall clock of the reference's loop through JNI-typed calls: GPU only)
    assert line["kernel_rate_isolated"] > 0 and line["value_one_frame_at_a_time"] is None and line["comm_cus_per_xcd"] == 0
