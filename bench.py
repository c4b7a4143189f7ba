#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X.

Metric (BASELINE.json): Mrays/s (primary + 1 bounce) at 1920x1080 on an 8192^3 SVO.
A "step" is one frame: every pixel's primary ray + its diffuse bounce (renderMode 0, the
reference's GI mode, svotrace.comp:443-560) through the HIP path, pool resident in HBM.
Rays = intersectOctree-equivalent casts actually performed (counted by an untimed
counting pass of the same frame); value = rays of all ranks / wall time of K steps.

N > 1 (one process per GPU, launched by torch.distributed.run).  The path shards by
screen tile: the pool is replicated by one RCCL broadcast, every rank renders every
N-th 8-pixel tile row (interleaved stripes, packed into one band of the gather buffer), and each step's bands are gathered to rank 0 over xGMI (RCCL
gather = one direct send per peer), overlapped with the next frame's traversal on a
second stream.  Default --scaling weak: the per-GPU band stays 1920x1080 and the frame
grows to 1920 x (1080*N) rows (same camera, denser rows), so per-GPU work is fixed.
--scaling strong splits the SAME 1920x1080 frame into N bands instead; a 1.5 ms frame is
then bounded by the longest single path (about 0.5 ms of dependent loads), see DESIGN.md.

Also on the JSON line: roofline (algorithmic bytes / HIP-event kernel time vs 8 TB/s HBM)
and cpu_baseline (the CPU oracle timed on a bounded pixel subsample of the same frame).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=8192, help="SVO resolution N (N^3 voxels)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--mode", type=int, default=0, help="renderMode: 0 = GI primary + bounce (metric), 2 = primary + shadow")
    ap.add_argument("--bounces", type=int, default=2, help="path segments in mode 0 (2 = primary + 1 bounce)")
    ap.add_argument("--camera", default="K1")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("SVO_BENCH_PIPELINE", "1")),
                    help="0 one thread per pixel, 1 persistent waves (default), 2 staged wavefront")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("SVO_BENCH_INFLIGHT", "3")),
                    help="frames in flight (streams x output buffers); the next frame fills the GPU while the "
                         "previous one drains its longest paths")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-oracle sample time (0 = skip)")
    ap.add_argument("--hits", type=int, default=0, help="also store 16-byte hit records per pixel")
    ap.add_argument("--deinterleave", type=int, default=0,
                    help="rank 0 also reorders the gathered stripe-major tile buffers into frame order each step")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the SVO hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    force_comm = os.environ.get("SVO_BENCH_FORCE_COMM", "0") == "1"  # exercise the RCCL path on one GPU
    if world > 1 or (force_comm and "RANK" in os.environ):
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    use_comm = world > 1 or (force_comm and dist.is_initialized())

    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    from svo_raytracer_amd.tiles import stripe_layout, gather_bands_to_root, deinterleave

    W, H = args.width, args.height
    cam = CAMERAS[args.camera]
    ctx = hiplib.HipContext(local_rank)

    # ---- scene: built once on rank 0, replicated by one RCCL broadcast -------------------
    t_build = time.time()
    pool = None
    if rank == 0:
        pool, sstats = scene.build_scene(args.size)
        nbytes = int(pool.size)
    if world > 1:
        nb = torch.tensor([nbytes if rank == 0 else 0], dtype=torch.int64, device="cuda")
        dist.broadcast(nb, 0)
        nbytes = int(nb.item())
        dpool = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        if rank == 0:
            dpool.copy_(torch.from_numpy(pool))
        dist.broadcast(dpool, 0)
        torch.cuda.synchronize()
        ctx.pool_upload_device(dpool.data_ptr(), nbytes)
        del dpool
        torch.cuda.empty_cache()
    else:
        ctx.pool_upload(pool)
    t_build = time.time() - t_build

    # ---- frame state -----------------------------------------------------------------------
    H_total = H * world if args.scaling == "weak" else H
    ctx.resize(W, H_total)
    ctx.set_camera(cam)
    ctx.set_params(2, args.mode, nbytes, 0, args.bounces, 0, 1)  # frameNumber 2 = first frame (Main.java:16,275)
    ctx.set_pipeline(args.pipeline)
    if args.pipeline == 1 and max(2 if use_comm else 1, args.inflight) > 1:
        ctx.set_tuning(10, 9)  # several frames in flight share the CUs: 10 persistent waves per CU and frame,
                               # rounds once 7/16 of the traversing lanes have stopped (swept on MI355X, tools/sweep*.sh)
    # rank r renders tile rows r, r + N, r + 2N, ... (interleaved: every rank sees the same mix of near and
    # far terrain) and stores them packed in its band of the gather buffer
    s_first, s_step, s_n, s_out0, rows_per_rank = stripe_layout(H_total, world, rank)
    hp = rows_per_rank * world  # padded height so that every rank's band has the same size
    nbuf = min(4, max(2 if use_comm else 1, args.inflight))  # frame k drains / is gathered while frame k+1 is traced
    # (the library keeps a ring of 4+ per-frame work-counter / queue sets, so at most 4 frames may be in flight)
    color = [torch.zeros((hp, W), dtype=torch.int32, device="cuda") for _ in range(nbuf)]
    depth = [torch.zeros((hp, W), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    hits = [torch.zeros((hp, W, 4), dtype=torch.int32, device="cuda") for _ in range(nbuf)] if args.hits else None
    scratch_c = torch.zeros((rows_per_rank, W), dtype=torch.int32, device="cuda") if use_comm and rank == 0 else None
    scratch_d = torch.zeros((rows_per_rank, W), dtype=torch.float32, device="cuda") if use_comm and rank == 0 else None
    ctx.set_hit_records(bool(args.hits))
    ctx.set_stripes(s_first, s_step, s_n, s_out0)
    main_stream = torch.cuda.current_stream()
    streams = [main_stream] + [torch.cuda.Stream() for _ in range(nbuf - 1)]
    for st_ in streams[1:]:
        st_.wait_stream(main_stream)
    stream = main_stream
    ctx.set_stream(stream.cuda_stream)
    comm_stream = torch.cuda.Stream() if use_comm else None
    gathered = [None] * nbuf  # event: the gather that last read buffer b has finished
    state = {"k": 0, "timing": False}
    launch_events = []        # (start, end) HIP events around every launch of the timed region, on its own stream

    def step():
        b = state["k"] % nbuf
        state["k"] += 1
        stream = streams[b]
        ctx.set_stream(stream.cuda_stream)
        if gathered[b] is not None:
            stream.wait_event(gathered[b])
        ctx.bind_outputs(color[b].data_ptr(), depth[b].data_ptr(), hits[b].data_ptr() if hits is not None else None)
        if state["timing"]:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            ctx.dispatch_async()
            e1.record(stream)
            launch_events.append((e0, e1))
        else:
            ctx.dispatch_async()
        if use_comm:
            done = torch.cuda.Event()
            done.record(stream)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(done)
                gather_bands_to_root(dist, color[b], rank, world, rows_per_rank, force=force_comm, scratch=scratch_c)
                gather_bands_to_root(dist, depth[b], rank, world, rows_per_rank, force=force_comm, scratch=scratch_d)
                if rank == 0 and args.deinterleave:  # stripe-major as gathered -> frame order, on the frame owner
                    frame_c = deinterleave(color[b], world, rows_per_rank, H_total)
                    frame_d = deinterleave(depth[b], world, rows_per_rank, H_total)
                    state["frame"] = (frame_c, frame_d)
                ev = torch.cuda.Event()
                ev.record(comm_stream)
                gathered[b] = ev

    ctx.bind_outputs(color[0].data_ptr(), depth[0].data_ptr(), hits[0].data_ptr() if hits is not None else None)

    def drain():
        torch.cuda.synchronize()
        ctx.set_stream(main_stream.cuda_stream)
        ctx.bind_outputs(color[0].data_ptr(), depth[0].data_ptr(), hits[0].data_ptr() if hits is not None else None)

    # ---- ray count of the frame (untimed counting pass; identical image) -------------------
    cstats = ctx.count_frame()
    counts = torch.tensor([cstats["rays"], cstats["iterations"], cstats["alg_bytes"], cstats["pixels"],
                           cstats["nan_rays"]], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(counts)
    rays, iters, alg_bytes, pixels, nan_rays = [int(v) for v in counts.tolist()]

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    state["timing"] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    state["timing"] = False
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- kernel time by HIP events on the dispatch stream (rank 0's band), one frame at a time ---
    kernel_ms = float(np.mean([a_.elapsed_time(b_) for a_, b_ in launch_events]))  # with nbuf launches in flight
    drain()
    kms = ctx.time_frames(2, max(5, min(args.steps, 30)))
    kernel_ms_isolated = float(np.mean(kms))
    out_bytes_px = 8 + (16 if args.hits else 0)
    my_alg = cstats["alg_bytes"] + cstats["pixels"] * out_bytes_px

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays * args.steps / elapsed / 1e6
        # `nbuf` launches share the GPU at any time, each for `kernel_ms`; the device-level rate the HBM
        # roofline is about is bytes per launch / (timed region / launches).  With one frame in flight the two
        # are the same number.
        achieved_per_launch = my_alg / (kernel_ms * 1e-3) / 1e9
        achieved = my_alg / (elapsed / args.steps) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_per_launch.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "%d_%dx%d_m%d_p%d" % (args.size, W, H, args.mode, args.pipeline)
                traffic = tj.get(key)
            except Exception:
                traffic = None
        line = {
            "metric": "Mrays/s (primary + 1 bounce) at 1920x1080, 8192^3 SVO",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "%d^3 procedural terrain SVO (seed 1, %d bytes), %dx%d, renderMode %d (%s), camera %s, "
                            "pipeline %d, %dx%d pixels of interleaved tile rows per GPU, %d GPU(s), gathered to rank 0" % (
                                args.size, nbytes, W, H_total, args.mode,
                                "primary + %d bounce" % (args.bounces - 1) if args.mode == 0 else "primary + shadow ray",
                                args.camera, args.pipeline, W, rows_per_rank, world),
                "rays_per_frame": rays, "iterations_per_ray": round(iters / max(rays, 1), 2),
                "alg_bytes_per_ray": round(alg_bytes / max(rays, 1), 1), "nan_rays": nan_rays,
                "scene_build_s": round(t_build, 1), "frames_in_flight": nbuf,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel_ms": round(kernel_ms, 4), "launches_in_flight": nbuf,
                "achieved_per_launch": round(achieved_per_launch, 2),
                "kernel_ms_isolated": round(kernel_ms_isolated, 4), "alg_bytes_per_launch": int(my_alg),
            },
        }
        # ---- CPU baseline: the oracle on a bounded subsample of the same frame, 1 thread ----
        if args.cpu_seconds > 0 and world == 1:
            from oracle import oracle
            probe = oracle.render(pool, W, H, cam, 2, args.mode, bounces=args.bounces, xstep=32, ystep=32,
                                  want_hits=False)
            t1 = time.perf_counter()
            probe = oracle.render(pool, W, H, cam, 2, args.mode, bounces=args.bounces, xstep=32, ystep=32,
                                  want_hits=False)
            dt = max(time.perf_counter() - t1, 1e-4)
            per_px = dt / max(probe["stats"]["pixels"], 1)
            want_px = args.cpu_seconds / per_px
            stepxy = max(1, int(np.ceil(np.sqrt(W * H / want_px))))
            crays, cpix, nframes, dt = 0, 0, 0, 0.0
            t1 = time.perf_counter()
            while dt < args.cpu_seconds * 0.8 and nframes < 64:
                smp = oracle.render(pool, W, H, cam, 2 + nframes, args.mode, bounces=args.bounces, xstep=stepxy,
                                    ystep=stepxy, want_hits=False)
                crays += smp["stats"]["rays"]
                cpix += smp["stats"]["pixels"]
                nframes += 1
                dt = time.perf_counter() - t1
            line["cpu_baseline"] = {
                "value": round(crays / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
                "sample": "every %d-th pixel in x and y of the same frame, %d pass(es) with frameNumber 2.. "
                          "(%d pixels, %d rays, %.1f s), single-threaded C oracle; host has %d cores" % (
                              stepxy, nframes, cpix, crays, dt, os.cpu_count() or 0),
            }
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
