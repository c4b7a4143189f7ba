#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X.

Metric (BASELINE.json): Mrays/s (primary + 1 bounce) at 1920x1080 on an 8192^3 SVO.
A "step" is one frame of the reference's loop (Main.updateEarly, Main.java:257-289): frameNumber is
pre-incremented every frame (first frame = 2), so every timed frame is a different frame (its bounce
directions change, svotrace.comp:486); every pixel's primary ray + its diffuse bounce (renderMode 0,
the reference's GI mode, svotrace.comp:443-560) go through the HIP path, pool resident in HBM.
Rays = intersectOctree-equivalent casts actually performed (untimed counting pass of the first and
the last timed frame); value = rays of all ranks / wall time of K steps.

After the timed region the frames still held by the ring (the last `inflight` timed frames, as
rendered with that many launches in flight) are read back and a pixel subsample of each is compared
bit-for-bit with the CPU oracle: the line carries "verified": true, or the run exits non-zero.

N > 1: `python bench.py --gpus N` starts N ranks itself (a child torch.distributed.run, before this
process touches the GPU); under torch.distributed.run it is one of the ranks.  The path shards by
screen tile: the pool is replicated by one RCCL broadcast, every rank renders every N-th 8-pixel tile
row of the SAME frame (--scaling strong, the default: SURVEY 8d "same frame, tile-split, wall time
including the gather") packed into its chunk of the gather buffer, and each frame's chunks (colour +
depth in one message) are gathered to rank 0 over xGMI (one direct send per peer), overlapped with
the next frames' traversal.  --scaling weak keeps 1920x1080 pixels per GPU instead (the frame grows
to 1920 x 1080*N rows of the same view).

Throughput configuration (all of it on the JSON line, and all of it behind the C ABI: svo_ring_*): --inflight 6 submissions in
flight, each in a ring slot with a stream of its own, each a batch of --batch 4 consecutive frames (one persistent launch
whose waves run from frame to frame, so the launch's tail is paid once per batch).  A step is still ONE frame: K steps = K frames, the last dispatch
a partial batch if need be.  `--inflight 1 --batch 1` is the reference's own loop, one frame at a time.

Presets: --config C2 | C3 (default, the metric) | C4 | C5 are BASELINE.json's configs.
Also on the JSON line: roofline (algorithmic bytes / HIP-event kernel time vs 8 TB/s HBM, plus the
instruction-issue figures that actually bind, from the committed PMC passes when they were taken on
exactly these kernel sources) and cpu_baseline (the CPU oracle on a bounded subsample).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS = 1024            # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9
# gfx950 issues wave64 VALU instructions in two classes (tools/calib_valu2.hip, profiles/round3_calib_valu2.txt; 6 waves per
# SIMD): add / sub / mul / fma / logic / mov / right shift / v_bitop3 one per ~2.35 cycles, compares, v_cndmask, min / max,
# bit-field, shift-and-add, count and packed-f32 ops one per ~4.3.  The traversal trip (svo_travloop2.h) is 42 of the first
# and 34 of the second: 3.22 cycles per instruction; the round code is taken to mix alike.
VALU_CYCLES = (42 * 2.35 + 34 * 4.3) / 76.0

# frames per dispatch when --batch is not given (the same for every number of GPUs, so that the scaling curve compares like
# with like): a launch needs ~1.5 M rays or more to amortise its tail, and a rank's share of a 1080p frame shrinks with N.
# 6 dispatches in flight x 4 frames (profiles/round3_experiments.txt): in a long run every shape from
# 4 x 5 to 8 x 4 is within 1.5 % (5.35-5.45 Grays/s at 200 steps); a short timed region -- a driver's `--steps 20 --warmup 5`
# -- is all start and drain, and there 6 x 4 (5.15 at 20 steps, 5.16 at 40) beats round 2's 4 x 5 (4.83, 4.99) because the
# whole region is submitted at once and ends in ONE tail.
DEFAULT_BATCH = {1: 4, 2: 4, 4: 4, 8: 4}
DEFAULT_INFLIGHT = 6

PRESETS = {
    # name: size, width, height, mode, bounces (path segments), mirror mask, spp
    "C2": dict(size=2048, width=1920, height=1080, mode=1, bounces=2, mirror=0, spp=1),
    "C3": dict(size=8192, width=1920, height=1080, mode=0, bounces=2, mirror=0, spp=1),
    "C4": dict(size=8192, width=3840, height=2160, mode=0, bounces=5, mirror=0b1000, spp=1),
    # C5 "64 spp accumulated GI" in the reference's own terms: the cross-frame accumulation of svotrace.comp:712-719 over
    # 64 frames (frameNumber 2..65 on a fresh image, Main.java:16,275); a step = one such 64-frame sequence (svo_set_sequence)
    "C5": dict(size=8192, width=1920, height=1080, mode=0, bounces=2, mirror=0, spp=1, seq=64),
    # the library's own reading of the shader's commented-out SAMPLES loop (:668-670): 64 samples per frame, seeds
    # frameNumber + sample, fp32 mean -- HIP <-> oracle only, no reference behaviour
    "C5spp": dict(size=8192, width=1920, height=1080, mode=0, bounces=2, mirror=0, spp=64, seq=1),
}
for _p in PRESETS.values():
    _p.setdefault("seq", 1)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(PRESETS), default=None, help="BASELINE.json config preset (default: C3)")
    ap.add_argument("--size", type=int, default=None, help="SVO resolution N (N^3 voxels)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--mode", type=int, default=None, help="renderMode: 0 = GI primary + bounce (metric), 2 = primary + shadow")
    ap.add_argument("--bounces", type=int, default=None, help="path segments in mode 0 (2 = primary + 1 bounce)")
    ap.add_argument("--mirror", type=lambda v: int(v, 0), default=None, help="bit mask of mirror materials (svotrace.comp:500-504)")
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel accumulated per frame (svotrace.comp:668-670)")
    ap.add_argument("--seq", type=int, default=None, help="frames of the cross-frame accumulation (svotrace.comp:712-719) per step: "
                                                          "one step = frameNumber 2 .. seq + 1 blended into one image (svo_set_sequence)")
    ap.add_argument("--camera", default="K1", help="K0 (the reference's default, Main.java:120), K1, K2 (grazing): SURVEY 8(d); CAVE (with --scene caves): "
                                                   "inside the scene's largest cave")
    ap.add_argument("--scene", choices=["terrain", "caves", "dust"], default="terrain",
                    help="scene family: terrain = the integer-noise height field (SURVEY 8d), built on the GPU from its two maps; caves = "
                         "the same terrain under levels of hashed balls that carve it or float over it (overhangs, cave mouths, boulders, "
                         "floating debris: scene/svo_scene.c family 1), built on the host cores and uploaded; dust = the terrain under a "
                         "field of floating particles (family 2: the hostile case of an octree walk, 200 - 400 iterations per ray)")
    ap.add_argument("--seed", type=int, default=1, help="seed of the scene's integer noise")
    ap.add_argument("--amp", type=int, default=8, help="terrain amplitude in sixteenths of an octave's cell (8 = the default terrain)")
    ap.add_argument("--dens", type=int, default=None, help="caves: probability / 256 that a cell next to a surface holds a ball (default 64); "
                                                           "dust: probability / 256 that an air cell of the dust level holds a particle (default 24)")
    ap.add_argument("--ref-loop", type=int, default=None, help="(default: as --default-abi) also time the reference's own loop -- per frame nSetCamera, nSetParams, nDispatchAsync, "
                                                            "nReadPixel at the crosshair, through JNI-typed calls, wall clock (value_one_frame_at_a_time; "
                                                            "one GPU, default pipeline)")
    ap.add_argument("--by-camera", type=int, default=None, help="(default: as --default-abi) also measure the default configuration from each of SURVEY 8(d)'s cameras "
                                                             "K0 / K1 / K2, 100 verified steps each (value_by_camera; one GPU, C3 only)")
    ap.add_argument("--camera-path", choices=["static", "orbit"], default="static",
                    help="orbit: every timed frame carries its own camera (Camera.rotate + strafe through the host mirror) and "
                         "frameNumber 1 (Main resets it on motion, Main.java:225-233, 275): svo_ring_submit_cams")
    ap.add_argument("--moving", type=int, default=1, help="also measure the default configuration with a moving camera "
                                                          "(value_moving_camera; one GPU, static runs only)")
    ap.add_argument("--long-steps", type=int, default=400, help="steps of the second, longer timed region (value_long_run; 0 = skip)")
    ap.add_argument("--default-abi", type=int, default=1, help="also run the configuration through JNI-typed calls only, with no tuning / "
                                                                "pipeline call (value_default_abi; one GPU, default config only)")
    ap.add_argument("--fallback", type=int, default=1, help="N > 1 without a launcher: when a rung (driver / exchange) fails, times out or "
                                                            "does not verify, go on with the next one in a fresh child process")
    ap.add_argument("--rung-timeout", type=float, default=480.0, help="N > 1 without a launcher: wall-clock limit of one rung, seconds")
    ap.add_argument("--probe", type=int, default=1, help="N > 1 under a launcher (torch.distributed.run) with the default exchange: before this "
                                                         "rank touches the GPU, a child process per rank runs a few small frames with the RCCL "
                                                         "gather; if any rank's child fails or times out, all ranks go on with the copy exchange")
    ap.add_argument("--probe-timeout", type=float, default=300.0, help="wall-clock limit of a probe child, seconds")
    ap.add_argument("--driver", choices=["torch", "group"], default="torch",
                    help="N > 1: torch = one process per GPU under torch.distributed (RCCL gather or IPC copies); group = ONE "
                         "process, the N GPUs behind the C ABI (svo_group_*: peer copies or RCCL send / receive inside the library)")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("SVO_BENCH_PIPELINE", "1")),
                    help="0 one thread per pixel, 1 persistent waves (default), 2 staged wavefront")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("SVO_BENCH_INFLIGHT", str(DEFAULT_INFLIGHT))),
                    help="frames in flight (streams x output buffers); the next frame fills the GPU while the "
                         "previous one drains its longest paths")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("SVO_BENCH_BATCH", "0")),
                    help="frames per dispatch (svo_set_batch): one persistent launch carries that many consecutive frames, so "
                         "its tail is paid once per batch; 0 = default (see DEFAULT_BATCH)")
    ap.add_argument("--waves", type=int, default=-1, help="persistent waves per CU and launch (svo_set_tuning); -1 = no call: the library's "
                                                          "own choice (10 for a ring of several slots, fill the GPU otherwise)")
    ap.add_argument("--thresh", type=int, default=-1, help="refill round threshold in sixteenths (svo_set_tuning); -1 = no call: the library's 9")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-oracle sample time (0 = skip)")
    ap.add_argument("--hits", type=int, default=0, help="also store 16-byte hit records per pixel")
    ap.add_argument("--verify", type=int, default=1, help="check the ring's frames against the CPU oracle after timing")
    ap.add_argument("--isolated", type=int, default=1, help="also time single frames with the GPU to themselves (kernel_ms_isolated); "
                                                          "0 in PMC passes, so that every launch of the kernel is a timed-region launch")
    ap.add_argument("--beam", type=int, default=0, help="useBeamOptimization (coarse depth pre-pass, Main.java:257-266)")
    ap.add_argument("--comm-cus", type=int, default=-1, help="CUs per XCD the render streams leave free for the collective's kernels "
                                                              "(svo_set_reserved_cus); -1 = 1 when ranks exchange tiles, else 0")
    ap.add_argument("--exchange", choices=["rccl", "copy"], default=os.environ.get("SVO_BENCH_EXCHANGE") or None,
                    help="how a rank's tiles reach rank 0: one RCCL gather per dispatch (default of --driver torch), or device-to-device "
                         "copies into rank 0's buffer (svo_ring_forward_slot: SDMA, no CU slots needed; default of --driver group, "
                         "whose RCCL exchange has never run between two devices)")
    ap.add_argument("--as-rank", default=None, help="r/n: render what rank r of n would, on one GPU, no communication")
    args = ap.parse_args(argv)
    args.exchange_given = args.exchange is not None
    for leg in ("ref_loop", "by_camera"):      # the legs behind the headline go together unless named
        if getattr(args, leg) is None:
            setattr(args, leg, args.default_abi)
    if args.exchange is None:
        args.exchange = "copy" if args.driver == "group" else "rccl"
    preset = PRESETS[args.config or "C3"]
    for k, v in preset.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


def usable_cores():
    """Cores this process may actually use: the affinity mask, cut by a cgroup CPU quota if there is one (os.cpu_count()
    reports the whole host even inside a limited container)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _strip_opts(argv, names):
    """argv without the options in `names` (each takes one value; `--opt v` and `--opt=v` forms)"""
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in names:
            skip = True
            continue
        if any(a.startswith(n + "=") for n in names):
            continue
        out.append(a)
    return out


def ladder(args):
    """The rungs `--gpus N` tries in order: the driver / exchange asked for (default: one process per GPU under
    torch.distributed with an RCCL gather), then the paths that need less of the node -- the same driver with device-to-device
    copies instead of the collective (no CU slots next to the persistent waves), then ONE process with the N GPUs behind
    the C ABI (svo_group_*, peer copies)."""
    first = (args.driver, args.exchange)
    if not args.fallback:
        return [first]
    order = [("torch", "rccl"), ("torch", "copy"), ("group", "copy")]
    # what comes after the rung asked for; never back to something that needs more of the node than what failed
    return [first] + (order[order.index(first) + 1:] if first in order else [("group", "copy")])


def rung_command(args, driver, exchange, argv):
    rest = _strip_opts(argv, ("--driver", "--exchange"))
    me = os.path.abspath(__file__)
    if driver == "torch":
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), me] + rest + ["--driver", "torch", "--exchange", exchange]
    return [sys.executable, me] + rest + ["--driver", "group", "--exchange", exchange]


def run_rung(cmd, timeout_s, env=None):
    """One rung = one FRESH child process (group), never a retry inside a process that has initialised HIP.  Returns
    (rc or None on timeout, the last JSON line of its stdout or None, the last line of its stderr)."""
    import signal
    import tempfile
    with tempfile.TemporaryFile("w+") as out, tempfile.TemporaryFile("w+") as err:
        p = subprocess.Popen(cmd, stdout=out, stderr=err, env=env, start_new_session=True)
        try:
            rc = p.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            rc = None
            for sig in (signal.SIGTERM, signal.SIGKILL):     # the process group this rung started, nothing else
                try:
                    os.killpg(p.pid, sig)
                except ProcessLookupError:
                    break
                try:
                    p.wait(timeout=10)
                    break
                except subprocess.TimeoutExpired:
                    continue
        out.seek(0)
        err.seek(0)
        line = None
        for ln in out.read().splitlines():
            ln = ln.strip()
            if ln.startswith("{") and ln.endswith("}"):
                try:
                    line = json.loads(ln)
                except Exception:
                    pass
        etxt = [ln for ln in err.read().splitlines() if ln.strip()]
        return rc, line, (etxt[-1][-300:] if etxt else "")


def launch_ranks(args, argv=None, runner=run_rung):
    """--gpus N > 1 without a launcher.  This process never touches the GPU (device_count() does not initialise it): it
    walks the ladder, one fresh child per rung with a wall-clock limit, and prints the line of the first rung that exits
    zero with a verified frame -- carrying `driver`, `exchange` and `fallback_from` (the rungs that failed before it, each
    with how it failed)."""
    argv = list(sys.argv[1:] if argv is None else argv)
    if runner is run_rung:
        import torch
        have = torch.cuda.device_count()
        if args.gpus > have and os.environ.get("SVO_BENCH_ONE_GPU", "0") != "1":   # (tests: every rank / member on GPU 0)
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, have))
    failed = []
    env = dict(os.environ, SVO_BENCH_CHILD="1")
    for driver, exchange in ladder(args):
        cmd = rung_command(args, driver, exchange, argv)
        # the copy exchange needs no collective on device memory: its ranks keep the control plane on gloo, so a node whose RCCL
        # failed the rung before does not fail this one the same way (probe_env does the same for the copy probe)
        renv = dict(env, SVO_BENCH_BACKEND="gloo") if (driver, exchange) == ("torch", "copy") else env
        rc, line, last_err = runner(cmd, args.rung_timeout, renv)
        ok = rc == 0 and line is not None and line.get("verified") is not False
        if ok:
            line["fallback_from"] = failed
            print(json.dumps(line), flush=True)
            return 0
        how = "timed out after %d s" % int(args.rung_timeout) if rc is None else \
            ("exit code %d" % rc if (rc != 0 or line is None) else "verified: false")
        failed.append({"driver": driver, "exchange": exchange, "failed": how, "stderr_tail": last_err})
        print("bench.py: rung driver=%s exchange=%s failed (%s): %s" % (driver, exchange, how, last_err), file=sys.stderr, flush=True)
    print(json.dumps({"metric": "Mrays/s (primary + 1 bounce) at 1920x1080, 8192^3 SVO", "value": None, "n_gpus": args.gpus,
                      "error": "every rung failed", "fallback_from": failed}), flush=True)
    return 1


KERNEL_SOURCES = ("svo_persistent.hip.h", "svo_travloop2.h", "svo_trav2.h", "svo_derive.hip.h", "svo_descword.h", "svo_travloop.h", "svo_trav.h",
                  "svo_device.h", "svo_fused.hip.h", "svo_kernels.h", "Makefile")
# (the comparators' sources -- csrc/variants/ -- are not in it: a run on them carries SVO_HIP_LIB / SVO_SPARE in its key, below)
# environment switches that select another kernel or another data path than the default's: PMC / stamps figures are keyed by them
KERNEL_ENV = ("SVO_SPARE", "SVO_RC_TABLE", "SVO_NORMAL_TABLE", "SVO_DERIVED", "SVO_FORCE_CAMS", "SVO_HIP_LIB")


def source_hash():
    """Hash of what the dominant kernel (persist_kernel) is compiled from, launch shape included: PMC figures taken on
    other sources are not reported."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "svo-raytracer_amd", "csrc")
    for n in KERNEL_SOURCES:
        h.update(n.encode())
        h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


def default_batch(args, world):
    """Frames per dispatch when --batch is not given: several samples per pixel already make one dispatch a long launch
    (the library folds them into it), so such frames go one per dispatch."""
    if args.spp > 1 or args.seq > 1:
        return 1
    return DEFAULT_BATCH.get(world, 4 if world > 8 else 1)


def pmc_key(args, width, height, nbuf, batch):
    """What a set of per-launch PMC figures belongs to (tools/pmc_pass.py writes under the same key)."""
    key = "%s_%dx%d_m%d_b%d_s%d_p%d_if%d" % (args.size, width, height, args.mode, args.bounces, args.spp, args.pipeline, nbuf)
    if args.seq > 1:
        key += "_seq%d" % args.seq
    if args.camera_path != "static":
        key += "_" + args.camera_path
    if batch > 1:
        key += "_B%d" % batch
    if args.beam:
        key += "_beam"
    if args.camera != "K1":
        key += "_" + args.camera
    if args.scene != "terrain":
        key += "_" + args.scene + ("_d%d" % args.dens if args.dens is not None else "")
    if args.seed != 1:
        key += "_seed%d" % args.seed
    if args.amp != 8:
        key += "_amp%d" % args.amp
    if args.mirror:
        key += "_mirror%x" % args.mirror
    for e in KERNEL_ENV:      # (a run on another kernel / library never picks up the default kernel's counters)
        v = os.environ.get(e)
        if v not in (None, ""):
            key += "_%s=%s" % (e, os.path.basename(v))
    return key


def pmc_for(key):
    """Per-launch PMC means (profiles/pmc_per_launch.json, written by tools/pmc_pass.py on the GPU box from separate
    rocprofv3 --pmc passes of this very command) -- only if they were taken on the current kernel sources."""
    p = os.path.join(ROOT, "profiles", "pmc_per_launch.json")
    try:
        j = json.load(open(p))
    except Exception:
        return None
    e = j.get(key)
    if not e or e.get("src_hash") != source_hash():
        return None
    return e


def run_default_abi(pool, W, H, cam, args, nbuf, batch, rays_per_frame, first_timed):
    """The configuration of the headline as a drop-in host drives it: a context of its own, JNI-typed calls only (the natives
    of integration/java/src/engine/HipRenderer.java, include/svo_hip_jni.h), and NO tuning / pipeline / hit-record call -- the
    launch shape is whatever the library picks for a ring of `nbuf` slots.  Same frame numbers, warm-up and step count as the
    timed region of `value`; then the same again over --long-steps; the last frame is checked against the CPU oracle."""
    import ctypes
    import numpy as np
    from svo_raytracer_amd import hiplib
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
    nRingCreate = fn("nRingCreate", jint, jlong, jint, jint, jint)
    nRingSubmit = fn("nRingSubmit", jint, jlong, jint, jint)
    nRingWait = fn("nRingWait", jint, jlong, jint)
    nRingReadColor = fn("nRingReadColor", jint, jlong, jint, jint, jlong)
    nRingReadDepth = fn("nRingReadDepth", jint, jlong, jint, jint, jlong)
    nDerivedInfo = fn("nDerivedInfo", jlong, jlong, jlong)
    nLaunchInfo = fn("nLaunchInfo", jint, jlong, jlong)

    def ok(rc, what):
        if rc < 0:
            raise RuntimeError("%s returned %d" % (what, rc))
        return rc

    j = nCreate(0)
    if j == 0:
        raise RuntimeError("nCreate returned 0")
    try:
        ok(nPoolUpload(j, pool.ctypes.data, pool.size), "nPoolUpload")
        ok(nSetCamera(j, *[float(v) for v in np.asarray(cam, np.float32).reshape(-1)]), "nSetCamera")
        ok(nResize(j, W, H), "nResize")
        ok(nRingCreate(j, nbuf, batch, 0), "nRingCreate")
        ok(nDerivedInfo(j, 0), "nDerivedInfo")     # the table resident before the timed region, as for `value`
        ok(nSetParams(j, 2, args.mode, int(pool.size), 0, args.bounces, args.mirror, 1), "nSetParams")
        state = {"frame": 2, "used": [False] * nbuf, "next": 0, "last": None}

        def run(n):
            while n > 0:
                k = min(batch, n)
                b = state["next"]
                if state["used"][b]:
                    ok(nRingWait(j, b), "nRingWait")    # awaitFrames(slot) before the slot's images are rendered over
                slot = ok(nRingSubmit(j, state["frame"], k), "nRingSubmit")
                state["used"][slot] = True
                state["last"] = (slot, k, state["frame"])
                state["next"] = (slot + 1) % nbuf
                state["frame"] += k
                n -= k

        def drain():
            for b in range(nbuf):
                if state["used"][b]:
                    ok(nRingWait(j, b), "nRingWait")

        def timed(n):
            # the region is bracketed like `value`'s: torch.cuda.synchronize() on both sides (a device-wide active wait;
            # six hipEventSynchronize in a row -- nRingWait per slot -- wake up late often enough to cost an 11 ms region 6 %:
            # profiles/round5_experiments.txt), the slots' own waits behind it for the bookkeeping
            import torch
            drain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(n)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            drain()
            return dt

        state["frame"] = 1002
        # the new context's pool copy and table touched, and the GPU back at its clocks after the CPU-side verification of the
        # legs before (`value`'s context had its counting passes for that): ~0.5 s of untimed frames.  (With 0.1 s an 11 ms
        # region of this leg read 6-8 % low and its 400-step region did not: profiles/round5_experiments.txt.)
        run(int(os.environ.get("SVO_ABI_PREWARM", "1000")))
        state["frame"] = 2     # then the frame numbers of `value`'s own warm-up and timed region
        run(args.warmup)
        el = timed(args.steps)
        wpc = ctypes.c_int32(0)
        waves = nLaunchInfo(j, ctypes.addressof(wpc))
        out = {"value": round(rays_per_frame * args.steps / el / 1e6, 2), "unit": "Mrays/s", "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(el / args.steps * 1e3, 4),
               "launch_shape": {"persistent_waves": int(waves), "waves_per_cu": int(wpc.value), "slots": nbuf, "frames_per_slot": batch},
               "what": "a second context driven through the JNI-typed exports only (nCreate, nPoolUpload, nSetCamera, nResize, "
                       "nRingCreate(%d, %d, 0), nSetParams, nRingSubmit / nRingWait): no nSetTuning, nSetPipeline or nSetHitRecords "
                       "call -- the library's defaults" % (nbuf, batch)}
        # the same region four more times: an 11 ms region's own spread on this box, next to its first reading
        out["value_repeats"] = [round(rays_per_frame * args.steps / timed(args.steps) / 1e6, 2) for _ in range(4)] if args.steps <= 40 else None
        if args.long_steps > 0:
            el2 = timed(args.long_steps)
            out["value_long_run"] = round(rays_per_frame * args.long_steps / el2 / 1e6, 2)
            out["long_run_steps"] = args.long_steps
        if args.verify:
            from oracle import oracle   # the checker; after the timed regions
            slot, k, first = state["last"]
            fr = first + k - 1
            rgba = np.zeros((H, W, 4), np.uint8)
            depth = np.zeros((H, W), np.float32)
            ok(nRingReadColor(j, slot, k - 1, rgba.ctypes.data), "nRingReadColor")
            ok(nRingReadDepth(j, slot, k - 1, depth.ctypes.data), "nRingReadDepth")
            step = 32
            bad, npx = 0, 0
            xs = np.arange(0, W, step)
            for y in range(0, H, step):
                ref = oracle.render(pool, W, H, cam, fr, args.mode, bounces=args.bounces, mirror_mask=args.mirror, spp=1,
                                    rows=(y, y + 1), xstep=step, want_hits=False)
                bad += int((rgba[y, xs] != ref["rgba"][y, xs]).any(axis=1).sum())
                bad += int((depth.view(np.uint32)[y, xs] != ref["depth"].view(np.uint32)[y, xs]).sum())
                npx += int(xs.size)
            out["verified"] = bad == 0 and npx > 0
            out["verification"] = "frame %d of the long run, every %d-th pixel in x and y (%d pixels) vs the CPU oracle: %d mismatches" % (fr, step, npx, bad)
        return out
    finally:
        nDestroy(j)


def run_reference_loop(pool, W, H, cam, args, nframes=300, warm=30):
    """The reference's own loop as the loop it is (Main.updateEarly, Main.java:132-146, 257-289): per frame the uniforms
    (nSetCamera, nSetParams), the dispatch (nDispatchAsync = glDispatchCompute + glMemoryBarrier, which return at once) and the
    crosshair read-back of THAT frame (nReadPixel at the image centre: Main reads it at the top of the next updateEarly, before
    anything else) -- through JNI-typed calls only, on a context of its own, timed by the wall clock around `nframes` frames and a
    final nSync.  Once with a static camera (frameNumber advances every frame) and once with one that moves every frame
    (frameNumber 1, Main.java:225-233, 275); rays of the counted frames through nCountFrame on the same context; the last frame
    of each leg verified against the CPU oracle.  Then the static leg again without the alternating image sets, and without
    sets and pick (= round 5's loop: a whole-frame wait per pick), for the A/B the header of svo_dispatch_async cites."""
    import ctypes
    import numpy as np
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS, orbit_path
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
    nReadColor, nReadDepth = fn("nReadColor", jint, jlong, jlong), fn("nReadDepth", jint, jlong, jlong)
    nSetHitRecords = fn("nSetHitRecords", jint, jlong, jint)
    nCountFrame = fn("nCountFrame", jint, jlong, jlong)
    nDerivedInfo = fn("nDerivedInfo", jlong, jlong, jlong)
    nPickInfo = fn("nPickInfo", jlong, jlong, jlong, jlong)
    nSetPick, nSetOverlap = fn("nSetPick", jint, jlong, jint, jint), fn("nSetOverlap", jint, jlong, jint)

    def ok(rc, what):
        if rc < 0:
            raise RuntimeError("%s returned %d" % (what, rc))
        return rc

    j = nCreate(0)
    if j == 0:
        raise RuntimeError("nCreate returned 0")
    try:
        ok(nPoolUpload(j, pool.ctypes.data, pool.size), "nPoolUpload")
        ok(nResize(j, W, H), "nResize")
        ok(nSetHitRecords(j, 1 if args.hits else 0), "nSetHitRecords")
        ok(nDerivedInfo(j, 0), "nDerivedInfo")
        if os.environ.get("SVO_LOOP_WAVES"):      # experiment knob (tools/loop_shape.py): the launch shape of the loop's dispatches
            ok(fn("nSetTuning", jint, jlong, jint, jint)(j, int(os.environ["SVO_LOOP_WAVES"]), int(os.environ.get("SVO_LOOP_THRESH", "0"))), "nSetTuning")
        if os.environ.get("SVO_LOOP_SETS"):       # experiment knob: image sets svo_dispatch_async takes turns on (2 .. 4)
            ok(nSetOverlap(j, int(os.environ["SVO_LOOP_SETS"])), "nSetOverlap")
        cx, cy = W // 2, H // 2
        one = np.zeros(1, np.float32)
        st = hiplib.Stats()

        def frames(cams, fnums, n, first):
            """n frames of the loop; frame i uses cams[i % len] and frame number fnums(i)"""
            for i in range(first, first + n):
                ok(nSetCamera(j, *cams[i % len(cams)]), "nSetCamera")
                ok(nSetParams(j, fnums(i), args.mode, int(pool.size), 0, args.bounces, args.mirror, 1), "nSetParams")
                ok(nDispatchAsync(j), "nDispatchAsync")
                ok(nReadPixel(j, cx, cy, 0, one.ctypes.data, 0), "nReadPixel")

        def leg(cams, fnums, what):
            camsf = [[float(v) for v in np.asarray(c, np.float32).reshape(-1)] for c in cams]
            rays = []
            for i in sorted({warm + (nframes - 1) * k // 4 for k in range(5)}):
                ok(nSetCamera(j, *camsf[i % len(camsf)]), "nSetCamera")
                ok(nSetParams(j, fnums(i), args.mode, int(pool.size), 0, args.bounces, args.mirror, 1), "nSetParams")
                ok(nCountFrame(j, ctypes.addressof(st)), "nCountFrame")
                rays.append(int(st.rays))
            frames(camsf, fnums, warm, 0)
            ok(nSync(j), "nSync")
            m0 = nPickInfo(j, 0, 0)
            t0 = time.perf_counter()
            frames(camsf, fnums, nframes, warm)
            ok(nSync(j), "nSync")
            dt = time.perf_counter() - t0
            early = nPickInfo(j, 0, 0) - m0
            r = {"value": round(float(np.mean(rays)) * nframes / dt / 1e6, 2), "unit": "Mrays/s", "frames": nframes,
                 "ms_per_frame": round(dt / nframes * 1e3, 4), "rays_per_frame": int(np.mean(rays)), "picks_answered_before_the_frame_ended": int(early),
                 "what": what}
            if args.verify:
                from oracle import oracle   # the checker; behind the timed region
                last = warm + nframes - 1
                rgba, depth = np.zeros((H, W, 4), np.uint8), np.zeros((H, W), np.float32)
                ok(nReadColor(j, rgba.ctypes.data), "nReadColor")
                ok(nReadDepth(j, depth.ctypes.data), "nReadDepth")
                step, bad, npx = 32, 0, 0
                xs = np.arange(0, W, step)
                for y in list(range(0, H, step)) + [cy]:
                    ref = oracle.render(pool, W, H, np.asarray(cams[last % len(cams)], np.float32), fnums(last), args.mode, bounces=args.bounces,
                                        mirror_mask=args.mirror, spp=1, rows=(y, y + 1), xstep=step if y != cy else 1, want_hits=False)
                    sel = xs if y != cy else np.array([cx])
                    bad += int((rgba[y, sel] != ref["rgba"][y, sel]).any(axis=1).sum())
                    bad += int((depth.view(np.uint32)[y, sel] != ref["depth"].view(np.uint32)[y, sel]).sum())
                    npx += int(sel.size)
                bad += int(one.view(np.uint32)[0] != depth.view(np.uint32)[cy, cx])      # the last pick = the image's crosshair pixel
                r["verified"] = bad == 0
                r["verification"] = "last frame (frameNumber %d): every %d-th pixel + the crosshair (%d pixels) and the last pick vs the CPU oracle: %d mismatches" % (
                    fnums(last), step, npx, bad)
            return r

        static = leg([cam], lambda i: 2 + i, "static camera, frameNumber 2, 3, ... (Main.java:275); the library's defaults: four image sets in turn, "
                                              "the pick answered by a one-wave launch of its own")
        mcams, _ = orbit_path(warm + nframes + 4, start=(args.camera if args.camera in CAMERAS else cam))
        moving = leg(list(mcams), lambda i: 1, "a camera that moves every frame (Camera.rotate + strafe through the host mirror), frameNumber 1 on every frame")
        ok(nSetOverlap(j, 2), "nSetOverlap")
        two_sets = leg([cam], lambda i: 2 + i, "static camera; TWO image sets in turn (svo_set_overlap 2: at most two frames in flight), pick from the mail")
        ok(nSetOverlap(j, 0), "nSetOverlap")
        pick_only = leg([cam], lambda i: 2 + i, "static camera; one stream and one image set (svo_set_overlap 0), pick from the mail")
        ok(nSetPick(j, -1, -1), "nSetPick")
        neither = leg([cam], lambda i: 2 + i, "static camera; one stream, one image set, no pick: every read-back waits for its frame (round 5's loop)")
        return {"static": static, "moving": moving, "two_image_sets": two_sets, "without_overlap": pick_only, "without_overlap_and_pick": neither,
                "image_sets": "library default (svo_set_overlap 1 = four sets in turn: up to four frames in flight, each launch 8 persistent waves per CU)",
                "calls_per_frame": "nSetCamera, nSetParams, nDispatchAsync, nReadPixel(%d, %d) -- JNI-typed exports, wall clock, a context of its own" % (cx, cy)}
    finally:
        nDestroy(j)


def stamps_for(key):
    """Mean lanes traversing per trip of the assembly loop (tools/stamps.py on an -DSVO_STAMPS=1 build of the same sources:
    profiles/stamps_per_launch.json) -- only if taken on the current kernel sources."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "stamps_per_launch.json")))
    except Exception:
        return None
    e = j.get(key)      # (no fall-back to another launch shape's entry: a line carries its own shape's figures or none)
    if not e or e.get("src_hash") != source_hash():
        return None
    return {k: v for k, v in e.items() if k not in ("src_hash", "all")}      # ("all": the other launch shapes of the same pass)


def probe_command(args, exchange):
    """A rank of the small run that tries an exchange out: 512^3, 640x360, 12 frames, verified against the oracle on rank 0."""
    return [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--size", "512", "--width", "640", "--height", "360",
            "--steps", "8", "--warmup", "4", "--inflight", "3", "--batch", "2", "--cpu-seconds", "0", "--moving", "0", "--default-abi", "0",
            "--long-steps", "0", "--isolated", "0", "--probe", "0", "--driver", "torch", "--exchange", exchange]


def probe_env(exchange, attempt, port=None):
    """The environment of a probe child: this rank's RANK / LOCAL_RANK / WORLD_SIZE, a rendezvous of its own (the launcher's
    store is the parent job's: the children meet through a TCP store rank 0's child opens at `port` -- a free port rank 0 of the
    parent job picked and the gloo group carried to every rank; MASTER_PORT + 101 + attempt without one)."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_") and k != "TORCH_NCCL_ASYNC_ERROR_HANDLING"}
    env["MASTER_PORT"] = str(port if port else int(os.environ.get("MASTER_PORT", "29500")) + 101 + attempt)
    env["SVO_BENCH_CHILD"] = "1"
    if exchange == "copy":
        env["SVO_BENCH_BACKEND"] = "gloo"      # the copy exchange needs no collective on device memory: control plane only
    return env


def negotiate_exchange(args, dist, torch, runner=run_rung):
    """N > 1 under a launcher, default exchange: decide -- before this process touches the GPU -- which exchange the ranks use.
    Every rank starts a probe CHILD (a rank of a small job of its own: a few frames with the RCCL gather next to CU-masked
    persistent launches, the combination that has never run between two devices); the ranks then agree over a CPU (gloo)
    group: the RCCL gather if every rank's child came back clean, else the same question for the copy exchange, which needs no
    collective.  Returns (exchange, backend, tried): backend "gloo" = the control plane stays on the CPU group made here."""
    tried = []
    dist.init_process_group("gloo")
    for attempt, exchange in enumerate(("rccl", "copy")):
        # a rendezvous port nobody holds: rank 0 asks the kernel for one, everybody learns it over the group that exists already
        pt = torch.tensor([_free_port() if dist.get_rank() == 0 else 0], dtype=torch.int32)
        dist.broadcast(pt, src=0)
        rc, line, err = runner(probe_command(args, exchange), args.probe_timeout, probe_env(exchange, attempt, int(pt.item())))
        mine = rc == 0 and (line is None or line.get("verified") is not False)
        t = torch.tensor([1 if mine else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(int(t.item()))
        if ok:
            break
        how = "timed out after %d s" % int(args.probe_timeout) if rc is None else ("exit code %s" % rc if rc != 0 else
                                                                                     ("verified: false" if not mine else "failed on another rank"))
        tried.append({"driver": "torch", "exchange": exchange, "failed": "probe: " + how, "stderr_tail": err})
    else:
        exchange = "copy"      # nothing came back clean: the path that needs the least, and let the run itself say what is wrong
    if exchange == "rccl":
        dist.destroy_process_group()      # the run makes its NCCL group as always
        return "rccl", "nccl", tried
    return exchange, "gloo", tried


def main(argv=None, ctx_factory=None):
    """ctx_factory: tests only -- a callable(local_rank) returning a CPU stand-in for hiplib.HipContext; the whole of main()
    then runs on CPU tensors under gloo (tests/test_bench_main_gloo.py), exercising the sharding, counting, timing
    protocol and the JSON line without a GPU.  The product path has no such fallback: without it a GPU is required."""
    args = parse(argv)
    stub = ctx_factory is not None
    dev = "cpu" if stub else "cuda"
    group_mode = args.driver == "group" and "RANK" not in os.environ
    if args.gpus > 1 and "RANK" not in os.environ and os.environ.get("SVO_BENCH_CHILD") != "1" and not stub:
        sys.exit(launch_ranks(args, argv))
    # the dispatches in flight, the gather and torch's own stream each want a hardware queue of their own; HIP's default
    # of 4 makes two of the frame streams share one (their launches then serialise).  Read when the runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    negotiated = []
    if (not stub and world > 1 and args.probe and not args.exchange_given and args.exchange == "rccl"
            and os.environ.get("SVO_BENCH_CHILD") != "1" and os.environ.get("SVO_BENCH_BACKEND", "") != "gloo"):
        # first contact with a node: try the exchange out in child processes before this one touches the GPU
        args.exchange, backend, negotiated = negotiate_exchange(args, dist, torch)
        if backend == "gloo":
            os.environ["SVO_BENCH_BACKEND"] = "gloo"
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the SVO hot path has no CPU fallback")

    def sync():
        if not stub:
            torch.cuda.synchronize()

    # SVO_BENCH_ONE_GPU=1 (tests): every rank on GPU 0 -- RCCL refuses that, the copy exchange (IPC) and gloo do not
    one_gpu = os.environ.get("SVO_BENCH_ONE_GPU", "0") == "1"
    if one_gpu:
        local_rank = 0
    if not stub:
        torch.cuda.set_device(local_rank)
    force_comm = os.environ.get("SVO_BENCH_FORCE_COMM", "0") == "1"  # exercise the RCCL path on one GPU
    if (world > 1 or (force_comm and "RANK" in os.environ)) and not dist.is_initialized():
        if stub or os.environ.get("SVO_BENCH_BACKEND", "") == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    use_comm = (world > 1 and args.exchange == "rccl") or (force_comm and dist.is_initialized())

    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    from svo_raytracer_amd.framering import FrameRing, GroupAsContext, GroupRing, replicate_pool

    W, H = args.width, args.height
    if args.camera == "CAVE":       # a camera inside the largest cave of the "caves" scene (cameras.cave_camera): not one of SURVEY 8(d)'s
        if args.scene != "caves":
            raise SystemExit("bench.py: --camera CAVE needs --scene caves")
        from svo_raytracer_amd.cameras import cave_camera
        cam = cave_camera(args.size, args.seed, args.amp, args.dens)
    else:
        cam = CAMERAS[args.camera]
    # ngpu: GPUs that share a frame -- the ranks of the torch driver, or the members of ONE process's group (svo_group_*)
    ngpu = args.gpus if group_mode else world
    group = None
    if group_mode:
        group = hiplib.HipGroup([0] * ngpu if one_gpu else list(range(ngpu)))
        ctx = GroupAsContext(group)
    else:
        ctx = ctx_factory(local_rank) if stub else hiplib.HipContext(local_rank)

    # ---- scene: built once on rank 0 (height / material maps on the host cores, the pool on the GPU by
    # svo_build_from_heightmap), replicated by one RCCL broadcast ------------------------------------------
    t_build = time.time()
    pool = None
    if rank == 0 and args.scene in ("caves", "dust"):
        # families 1 and 2 have no height map to build from: the host cores make the pool, svo_pool_upload copies it (Renderer.addSSBO)
        dens = (scene.CAVES_DENS if args.scene == "caves" else scene.DUST_DENS) if args.dens is None else args.dens
        # (tools/matrix.py runs several cells on one scene: SVO_SCENE_CACHE = a directory that keeps the pool between runs)
        cached = os.environ.get("SVO_SCENE_CACHE") and os.path.join(os.environ["SVO_SCENE_CACHE"],
                                                                    "%s_%d_s%d_a%d_d%d.npy" % (args.scene, args.size, args.seed, args.amp, dens))
        if cached and os.path.exists(cached):
            pool = np.load(cached)
        else:
            pool, _ = scene.build_scene3(args.size, args.seed, args.amp, dens if args.scene == "caves" else 0, dens if args.scene == "dust" else 0)
            if cached:
                np.save(cached, pool)
        nbytes = int(pool.size)
        ctx.pool_upload(pool)
        t_build = time.time() - t_build
    elif rank == 0:
        hmap, mmap = scene.scene_maps(args.size, args.seed, args.amp)
        nbytes = ctx.build_from_heightmap(hmap, mmap)
        t_build = time.time() - t_build
        del hmap, mmap
        pool = ctx.pool_download(nbytes)     # host copy: source of the broadcast, and what the oracle checks against
    if world > 1:
        dpool = replicate_pool(dist, pool, rank, world, device=dev)
        nbytes = int(dpool.numel())
        sync()
        if rank != 0:
            ctx.pool_upload_device(dpool.data_ptr(), nbytes)
        del dpool
        if not stub:
            torch.cuda.empty_cache()
    if rank != 0:
        t_build = time.time() - t_build

    # ---- frame state -----------------------------------------------------------------------------
    H_total = H * ngpu if args.scaling == "weak" else H
    as_rank = tuple(int(v) for v in args.as_rank.split("/")) if args.as_rank else None
    ctx.resize(W, H_total)
    ctx.set_camera(cam)
    ctx.set_pipeline(args.pipeline)
    nbuf = min(8, max(2 if (use_comm or world > 1) else 1, args.inflight))
    batch = args.batch if args.batch > 0 else default_batch(args, ngpu if as_rank is None else as_rank[1])
    if args.seq > 1:
        batch = 1      # a step is a whole sequence: one submission, one image
    # The launch shape is the library's own (include/svo_hip.h, svo_set_tuning: a ring of several slots runs 10 persistent
    # waves per CU and launch, a round once at most 9/16 of the lanes are still traversing; one launch at a time fills the GPU):
    # what a drop-in host gets without any call is what is measured here.  --waves / --thresh are experiment knobs.
    if args.pipeline == 1 and (args.waves >= 0 or args.thresh >= 0):
        ctx.set_tuning(max(args.waves, 0), max(args.thresh, 0))
    ctx.set_hit_records(bool(args.hits))
    if args.pipeline == 1 and hasattr(ctx, "derived_info"):
        ctx.derived_info()   # the interior-descriptor table is part of what is resident in HBM before the timed region starts
                             # (built at the first dispatch after a pool change otherwise: 6.7 ms at 8192^3, inside step 1 of a
                             # run without warm-up)
    comm_cus = args.comm_cus if args.comm_cus >= 0 else (1 if use_comm else 0)
    ctx.set_reserved_cus(comm_cus)
    params = dict(render_mode=args.mode, buffer_end=nbytes, use_beam=args.beam, bounces=args.bounces,
                  mirror_mask=args.mirror, spp=args.spp)
    if group_mode:
        ctx.set_params(2, args.mode, nbytes, args.beam, args.bounces, args.mirror, args.spp)
        ring = GroupRing(group, W, H_total, nbuf=nbuf, want_hits=bool(args.hits), first_frame=2, batch=batch,
                         exchange=args.exchange, advance=args.seq == 1)
    else:
        ring = FrameRing(ctx, W, H_total, world=world, rank=rank, nbuf=nbuf, device=dev,
                         dist=dist if (use_comm or world > 1) else None, want_hits=bool(args.hits), force_comm=force_comm,
                         first_frame=2, params=params, as_rank=as_rank, batch=batch, exchange=args.exchange,
                         advance=args.seq == 1)
    # a camera that moves: every frame of the run carries its own camera and frameNumber (svo_ring_submit_cams)
    path = None
    if args.camera_path == "orbit":
        from svo_raytracer_amd.cameras import orbit_path
        path = orbit_path(args.warmup + args.steps + 8, start=(args.camera if args.camera in CAMERAS else cam))

    # ---- ray count (untimed counting pass of the first and the last timed frame) ---------------------
    def count(frame):    # on the context's own stream and images (the ring's slots are not involved)
        ctx.set_batch(1, 0)
        ctx.set_params(frame, args.mode, nbytes, args.beam, args.bounces, args.mirror, args.spp)
        return ctx.count_frame()

    first_timed = 2 + args.warmup
    last_timed = first_timed + args.steps - 1
    if args.seq > 1:
        # a step renders frameNumber 2 .. seq + 1: its rays are those of all of them (same primaries, new bounces per frame)
        cs = [count(fr) for fr in range(2, 2 + args.seq)]
        keys = ("rays", "iterations", "alg_bytes", "pixels", "nan_rays")
        mine = [float(sum(c[k] for c in cs)) for k in keys]
    elif path is not None:
        # every timed frame has its own camera: count nine of them (all of a short run)
        idx = list(range(args.warmup, args.warmup + args.steps)) if args.steps <= 40 else \
            sorted({args.warmup + (args.steps - 1) * i // 8 for i in range(9)})
        cs = []
        for i in idx:
            ctx.set_camera(path[0][i])
            cs.append(count(int(path[1][i])))
        ctx.set_camera(cam)
        keys = ("rays", "iterations", "alg_bytes", "pixels", "nan_rays")
        mine = [sum(c[k] for c in cs) / float(len(cs)) for k in keys]
    else:
        # every timed frame of a short run; of a long one, nine frames spread over it (first and last included)
        counted = list(range(first_timed, last_timed + 1)) if args.steps <= 40 else \
            sorted({first_timed + (args.steps - 1) * i // 8 for i in range(9)})
        cs = [count(fr) for fr in counted]
        keys = ("rays", "iterations", "alg_bytes", "pixels", "nan_rays")
        mine = [sum(c[k] for c in cs) / float(len(cs)) for k in keys]
    counts = torch.tensor(mine, dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(counts)
    rays, iters, alg_bytes, pixels, nan_rays = [float(v) for v in counts.tolist()]
    if args.seq > 1:
        ctx.set_progressive(True)
        ctx.set_sequence(args.seq, True)
    if path is not None:
        ring.start_path(*path)

    def run_frames(n):      # exactly n frames: whole batches, then a partial one
        while n > 0:
            k = min(batch, n)
            ring.step(k)
            n -= k

    def timed(nsteps):
        """exactly nsteps steps between barrier + synchronize on both sides; returns this rank's seconds"""
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        run_frames(nsteps)
        if group_mode:
            ring.drain()      # every member's device, not only the current one
        sync()
        if world > 1:
            dist.barrier()
        sync()
        return time.perf_counter() - t0

    run_frames(args.warmup)
    ring.timing = True
    elapsed = timed(args.steps)
    ring.timing = False
    rank_ms = [elapsed / args.steps * 1e3]     # every rank's own ms per step: min / max go on the line next to the MAX that counts
    if world > 1:
        tall = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([elapsed], dtype=torch.float64, device=dev))
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in tall]
        elapsed = max(float(t.item()) for t in tall)
    # A driver's 20-step region is 11 ms at N = 1 and 1.5 ms of work per rank at N = 8 -- all start and drain, and millisecond
    # stalls of a box do not average out.  The same protocol again over a longer region, as an extra key.
    long_run = None
    if args.long_steps > 0 and args.seq == 1 and as_rank is None:
        ring.drain()
        el = timed(args.long_steps)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        long_run = (args.long_steps, el)

    # ---- the frames the ring still holds: the last nbuf timed frames, rendered with nbuf launches in flight ----
    ring.drain()

    def verify_ring(first_ok, vcam=None):
        """Every dispatch the ring still holds (its first and its last frame), a pixel subsample of each, bit for bit against
        the CPU oracle -- with the frame's own camera on a camera path, and through the oracle's statement of the cross-frame
        accumulation (svotrace.comp:712-719) for progressive sequences.  Returns (ok, description)."""
        from oracle import oracle   # the checker; never on the timed path
        step = 16 if W * H_total <= 1920 * 1080 * 2 else 32
        if args.spp > 8 or args.seq > 1:
            step *= 4
        rows_ok = ring.rendered_rows_mask().numpy()
        bad, npx, frames = 0, 0, []
        held = sorted({(b, k) for b in range(nbuf) if ring.frame_of[b] is not None for k in (0, ring.count_of[b] - 1)
                       if ring.cams_of[b] is not None or ring.frame_number(b, k) >= first_ok})
        for b, k in held:
            imgs = ring.frame_images(b, k)
            fr = imgs[0]
            fcam = ring.cams_of[b][k] if ring.cams_of[b] is not None else (cam if vcam is None else vcam)
            col = imgs[1].cpu().numpy().view(np.uint8).reshape(H_total, W, 4)
            dep = imgs[2].cpu().numpy()
            ys = np.flatnonzero(rows_ok)[::step]      # every step-th row of those this run rendered
            xs = np.arange(0, W, step)
            for y in ys:                               # the oracle on exactly those rows
                kw = dict(bounces=args.bounces, mirror_mask=args.mirror, spp=args.spp, rows=(int(y), int(y) + 1), xstep=step,
                          want_hits=False)
                if args.seq > 1:
                    last = np.zeros((H_total, W, 4), dtype=np.uint8)    # fresh: the image glTexStorage2D left
                    for f in range(fr, fr + args.seq):
                        ref = oracle.render(pool, W, H_total, fcam, f, args.mode, last_rgba=last, **kw)
                        last = ref["rgba"]
                else:
                    ref = oracle.render(pool, W, H_total, fcam, fr, args.mode, **kw)
                bad += int((col[y, xs] != ref["rgba"][y, xs]).any(axis=1).sum())
                bad += int((dep.view(np.uint32)[y, xs] != ref["depth"].view(np.uint32)[y, xs]).sum())
            npx += int(ys.size * xs.size)
            frames.append(int(fr))
        what = "frames %s" % frames if args.seq == 1 else "%d-frame progressive sequences starting at frame %s" % (args.seq, frames)
        info = "%s as left by the timed region (%d dispatches in flight x %d frames%s), every %d-th pixel in x and y " \
               "(%d pixels): rgba8 + depth bits vs the CPU oracle, %d mismatches" % (
                   what, nbuf, batch, ", each frame with its own camera" if ring.cams_of[held[0][0]] is not None else "",
                   step, npx, bad) if held else "nothing held"
        return bad == 0 and npx > 0, info

    verified, vinfo = None, None
    if args.verify and rank == 0:
        verified, vinfo = verify_ring(first_timed if args.seq == 1 else 2)

    # ---- the same configuration with a camera that moves (what a drop-in user of the reference's loop gets): every frame its
    # own camera (Camera.rotate + strafe through the host mirror) and frameNumber 1, 4 per launch, 6 launches in flight
    moving = None
    if (args.moving and path is None and args.seq == 1 and world == 1 and not group_mode and as_rank is None and not stub
            and args.pipeline == 1 and not args.beam):
        # (a leg behind the headline: whatever goes wrong here -- the host mirror missing, a build problem -- is reported on
        # the line, it must not lose the region already measured; a pixel mismatch still fails `verified`)
        try:
            from svo_raytracer_amd.cameras import orbit_path
            msteps = max(args.steps, 4 * batch * nbuf)
            mpath = orbit_path(msteps + 4 * batch * nbuf, start=(args.camera if args.camera in CAMERAS else cam))
            midx = sorted({2 * batch * nbuf + (msteps - 1) * i // 8 for i in range(9)})
            mrays = []
            for i in midx:
                ctx.set_camera(mpath[0][i])
                mrays.append(count(int(mpath[1][i]))["rays"])
            ctx.set_camera(cam)
            ring.start_path(*mpath)
            run_frames(2 * batch * nbuf)
            mel = timed(msteps)
            ring.drain()
            mok, minfo = verify_ring(0) if args.verify else (None, None)
            ring.start_path(None, None)
            moving = {"value": round(float(np.mean(mrays)) * msteps / mel / 1e6, 2), "unit": "Mrays/s", "steps": msteps,
                      "ms_per_step": round(mel / msteps * 1e3, 4), "rays_per_frame": int(np.mean(mrays)), "verified": mok,
                      "what": "camera path 'orbit' (Camera.rotate(0, 0.004, 0) + strafe per frame, frameNumber 1 on every frame as Main resets "
                              "it on motion), %d frames per launch with their own cameras (svo_ring_submit_cams), %d launches in flight; %s" % (
                                  batch, nbuf, minfo)}
            if args.verify and not mok:
                verified = False
        except Exception as e:     # noqa: BLE001
            moving = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
            try:
                ctx.set_camera(cam)
                ring.start_path(None, None)
                ring.drain()
            except Exception:
                pass

    # ---- the same configuration from each of SURVEY 8(d)'s three cameras: K0 = the reference's default (Main.java:120,
    # Camera.java:13-18), K1 = inside the cube, pitched toward the terrain, K2 = grazing.  100 steps each, same ring, same
    # protocol, the frames the region left in the ring verified against the oracle with that camera.
    by_camera = None
    if (args.by_camera and path is None and args.seq == 1 and world == 1 and not group_mode and as_rank is None and not stub
            and args.pipeline == 1 and args.spp == 1 and (args.config or "C3") == "C3"):
        by_camera = {}
        for cname in ("K0", "K1", "K2"):
            try:
                ring.drain()
                ctx.set_camera(CAMERAS[cname])
                f0 = ring.first_frame + ring.k
                csteps = 100
                crays = float(np.mean([count(f0 + 2 * batch * nbuf + i)["rays"] for i in (0, csteps // 2, csteps - 1)]))
                run_frames(2 * batch * nbuf)
                cel = timed(csteps)
                ring.drain()
                cok, cinfo = verify_ring(f0 + 2 * batch * nbuf, vcam=CAMERAS[cname]) if args.verify else (None, None)
                by_camera[cname] = {"value": round(crays * csteps / cel / 1e6, 2), "ms_per_step": round(cel / csteps * 1e3, 4),
                                    "rays_per_frame": int(crays), "steps": csteps, "verified": cok, "verification": cinfo}
                if args.verify and not cok:
                    verified = False
            except Exception as e:     # noqa: BLE001  (a leg behind the headline: reported, never loses the line)
                by_camera[cname] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
        try:
            ring.drain()
            ctx.set_camera(cam)
        except Exception:
            pass

    # ---- the same configuration as a drop-in host drives it: JNI-typed calls only (what HipRenderer.java's natives are), a
    # context of its own, NO tuning / pipeline / hit-record call -- createFrameRing(6, 4), submitFrames, awaitFrames
    default_abi = None
    if (args.default_abi and path is None and args.seq == 1 and ngpu == 1 and as_rank is None and not stub and not group_mode
            and args.pipeline == 1 and not args.beam and args.spp == 1 and rank == 0):
        try:
            default_abi = run_default_abi(pool, W, H_total, cam, args, nbuf, batch, rays, first_timed)
            if args.verify and default_abi.get("verified") is False:
                verified = False
        except Exception as e:     # noqa: BLE001
            default_abi = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}

    # ---- the reference's own loop, one frame at a time, as the loop it is: wall clock through JNI-typed calls
    ref_loop = None
    if (args.ref_loop and path is None and args.seq == 1 and ngpu == 1 and as_rank is None and not stub and not group_mode
            and args.pipeline == 1 and not args.beam and args.spp == 1 and rank == 0):
        try:
            ref_loop = run_reference_loop(pool, W, H_total, cam, args)
            if args.verify and any(v.get("verified") is False for v in ref_loop.values() if isinstance(v, dict)):
                verified = False
        except Exception as e:     # noqa: BLE001
            ref_loop = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- kernel time by HIP events on the dispatch streams; then one frame at a time with the GPU to itself ----
    # (svo_ring_query: events around every submission on its slot's stream; whole batches only, so that every launch
    # averaged carries the same number of frames)
    full = [m for m, n in ring.launch_ms if m > 0 and n == batch] or [m for m, n in ring.launch_ms if m > 0]
    kernel_ms = float(np.mean(full)) if full else 0.0   # with nbuf launches in flight
    if args.pipeline == 1 and args.waves > 0:
        ctx.set_tuning(0, max(args.thresh, 0))   # one frame at a time: back to the library's choice (fill the GPU)
    ctx.set_batch(1, 0)
    ctx.set_params(first_timed, args.mode, nbytes, args.beam, args.bounces, args.mirror, args.spp)
    # (the median: one frame in a few hundred takes milliseconds longer on these boxes, and a mean over 20 would carry it)
    kernel_ms_isolated = float(np.median(ctx.time_frames(2, max(5, min(args.steps, 30))))) if (args.isolated and not group_mode) else None
    out_bytes_px = 8 + (16 if args.hits else 0)
    my_alg = mine[2] + mine[3] * out_bytes_px

    rc = 0
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays * args.steps / elapsed / 1e6
        # `nbuf` launches share the GPU at any time; the device-level rate the HBM roofline is about is bytes per frame /
        # time per frame.  With one frame in flight that is bytes per launch / kernel time.
        achieved = my_alg / (elapsed / args.steps) / 1e9
        key = pmc_key(args, W, H_total, nbuf, batch)
        pmc = pmc_for(key) if ngpu == 1 and as_rank is None else None
        roof = {
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": (int((pmc["fetch_size_kb"] * 1024 * 2 + pmc["write_size_kb"] * 1024) / batch) if pmc else None),
            "traffic_note": "per frame: (FETCH_SIZE x 2 + WRITE_SIZE) KB of separate rocprofv3 --pmc passes; the x 2 is the scattered-load "
                            "correction of profiles/r01_fetch_size_calibration.txt (gfx950 tallies a 128-byte request as 64)",
            "frac_long_run": (round(my_alg / (long_run[1] / long_run[0]) / 1e9 / HBM_PEAK_GBS, 5) if long_run else None),
            "kernel_ms": round(kernel_ms, 4), "launches_in_flight": nbuf, "frames_per_launch": batch,
            "kernel_ms_isolated": round(kernel_ms_isolated, 4) if kernel_ms_isolated is not None else None,
            "alg_bytes_per_launch": int(my_alg * batch),
            "note": "a launch carries %d frame(s) and %d launches overlap, so kernel_ms > ms_per_step (= one frame); achieved = "
                    "bytes per frame / ms_per_step (device level); kernel_ms_isolated = one frame at a time, GPU filled by one launch.  "
                    "kernel_ms is taken between HIP events around the launch on its slot's stream (svo_ring_query): with more launches "
                    "in flight than fit the CUs at once it includes the launch's wait for CU slots, which rocprofv3's kernel duration "
                    "(first wave to last, profiles/) does not; one launch at a time the two agree" % (batch, nbuf),
        }
        if pmc:
            # the roof that binds: vector instruction issue.  Counters first (per launch of the PMC passes, hash-gated), the
            # class-weighted issue model next to them.
            o = pmc.get("other", {})
            br = {
                "kind": "valu-issue",
                "valu_insts_per_frame": int(pmc["sq_insts_valu"] / batch),
                "valu_lane_util": round(pmc["sq_thread_cycles_valu"] / (64.0 * pmc["sq_active_inst_valu"]), 4),
                "src_hash": pmc["src_hash"], "from": "profiles/pmc_per_launch.json",
            }
            if o.get("GRBM_GUI_ACTIVE") and o.get("SQ_WAVE_CYCLES"):
                # SQ_ACTIVE_INST_VALU counts quad-cycles a SIMD's VALU is busy: x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
                br["valu_busy_frac_counters"] = round(pmc["sq_active_inst_valu"] * 4.0 / (o["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS), 4)
                br["wait_inst_any_per_wave_cycle"] = round(o.get("SQ_WAIT_INST_ANY", 0.0) / o["SQ_WAVE_CYCLES"], 4)
                br["waves_per_simd_in_pmc_pass"] = round(o["SQ_WAVE_CYCLES"] * 4.0 / (o["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS), 3)
                br["counters_note"] = ("valu_busy_frac_counters = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), the SQ counters read "
                                       "as quad-cycles, measured with the PMC passes' own occupancy; wait_inst_any_per_wave_cycle = "
                                       "SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; waves_per_simd_in_pmc_pass = SQ_WAVE_CYCLES x 4 / (GRBM_GUI_ACTIVE / 8 x 1024): counter "
                                       "passes serialise launches, so they run at a lower occupancy than the timed region's 6 waves per SIMD")
            br["model"] = {
                "valu_issue_frac": round(pmc["sq_insts_valu"] / batch * VALU_CYCLES / (SIMDS * CLOCK_HZ * ms_per_step * 1e-3), 4),
                "cycles_per_valu_inst": round(VALU_CYCLES, 3),
                "note": "a MODEL with a fitted constant, not a counter: instructions x class-weighted issue cycles (calibrated on isolated "
                        "instruction kinds, tools/calib_valu2.hip) / (1024 SIMDs x 2.4 GHz x time); ~1 = the vector issue pipe is saturated",
            }
            st = stamps_for(key)
            if st:
                br["lanes_traversing_per_trip"] = st
            roof["binding_roof"] = br
        stripes = "%d GPU(s) x interleaved tile rows (%d pixel rows each), gathered to rank 0" % (ngpu, ring.rows_per_rank)
        if group_mode:
            stripes = "%d GPU(s) in ONE process behind the C ABI (svo_group_*), interleaved tile rows (%d pixel rows each), %s to member 0" % (
                ngpu, ring.rows_per_rank, "RCCL send / receive" if args.exchange == "rccl" else "peer copies")
        if as_rank:
            stripes = "what-if: the stripes of rank %d of %d on one GPU, no communication" % as_rank
        # the same frame through the reference's own loop -- one dispatch, then the crosshair read-back, then the next
        # (Main.updateEarly, Main.java:132-146, 257-289): what `value` would be without frames in flight
        # value_one_frame_at_a_time = that loop's wall clock (static camera; reference_loop has the moving camera and the A/B
        # legs); kernel_rate_isolated = rays / HIP-event time of back-to-back single-frame launches on one stream (what rounds
        # 1-5 printed under the first name: no read-back, no host in the loop)
        kernel_rate_isolated = (rays / (kernel_ms_isolated * 1e-3) / 1e6) if (kernel_ms_isolated and ngpu == 1 and as_rank is None and path is None) else None
        one_at_a_time = ref_loop["static"]["value"] if (ref_loop and "static" in ref_loop) else None
        line = {
            "metric": "Mrays/s (primary + 1 bounce) at 1920x1080, 8192^3 SVO",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": ngpu, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "verified": verified,
            "frames_in_flight": nbuf * batch,
            "value_one_frame_at_a_time": round(one_at_a_time, 2) if one_at_a_time else None,
            "value_one_frame_at_a_time_moving_camera": (ref_loop["moving"]["value"] if (ref_loop and "moving" in ref_loop) else None),
            "reference_loop": ref_loop,
            "kernel_rate_isolated": round(kernel_rate_isolated, 2) if kernel_rate_isolated else None,
            # the default configuration with a camera that moves every frame (None where it was not measured)
            "value_moving_camera": moving["value"] if moving else None,
            "moving_camera": moving,
            # the same configuration through JNI-typed calls only, no tuning / pipeline call: what a drop-in host gets by default
            "value_default_abi": default_abi["value"] if default_abi else None,
            "default_abi": default_abi,
            # the default configuration from each of SURVEY 8(d)'s cameras (100 verified steps each; None where not measured)
            "value_by_camera": ({k: v.get("value") for k, v in by_camera.items()} if by_camera else None),
            "by_camera": by_camera,
            # the same protocol over a longer region (a 20-step region is 11 ms at N = 1, ~1.5 ms of work per rank at N = 8)
            "value_long_run": (round(rays * long_run[0] / long_run[1] / 1e6, 2) if long_run else None),
            "long_run_steps": (long_run[0] if long_run else None),
            "driver": ("group: one process, svo_group_*" if group_mode else "torch.distributed: one process per GPU") if ngpu > 1 else None,
            "ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,
            "rank_ms_per_step": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4)},
            "gather_ms": ring.gather_ms(),
            "comm_cus_per_xcd": comm_cus, "exchange": (args.exchange if ngpu > 1 else None), "fallback_from": negotiated,
            "config": {
                "workload": "%s: %d^3 procedural %s SVO (seed %d, amplitude %d/16, %d bytes), %dx%d, renderMode %d (%s), %s, camera %s, "
                            "%s, pipeline %d, %s" % (
                                args.config or "C3", args.size,
                                "terrain" if args.scene == "terrain" else
                                ("caves (terrain + %d/256 hashed balls: overhangs, cave mouths, debris)" % (scene.CAVES_DENS if args.dens is None else args.dens)
                                 if args.scene == "caves" else
                                 "dust (terrain + floating particles in %d/256 of the air cells of edge N/256)" % (scene.DUST_DENS if args.dens is None else args.dens)),
                                args.seed, args.amp, nbytes, W, H_total, args.mode,
                                ("primary + %d bounce(s)%s" % (args.bounces - 1, ", mirror mask 0x%x" % args.mirror if args.mirror else ""))
                                if args.mode == 0 else ("primary + shadow ray" if args.mode == 2 else "primary only"),
                                ("%d spp" % args.spp) if args.seq == 1 else
                                ("%d accumulated frames per step = the reference's cross-frame accumulation (svotrace.comp:712-719) of frameNumber "
                                 "2..%d on a fresh image, one persistent launch per step (svo_set_sequence)" % (args.seq, args.seq + 1)),
                                args.camera if path is None else args.camera + " moving along path 'orbit' (a new camera every frame)",
                                ("frameNumber %d..%d (one per step)" % (first_timed, last_timed)) if (args.seq == 1 and path is None) else
                                ("frameNumber 1 on every frame (reset by the motion)" if path is not None else "every step the same sequence"),
                                args.pipeline, stripes),
                "rays_per_frame": int(round(rays / max(args.seq, 1))), "rays_per_step": int(round(rays)), "iterations_per_ray": round(iters / max(rays, 1), 2),
                "alg_bytes_per_ray": round(alg_bytes / max(rays, 1), 1), "nan_rays": int(round(nan_rays)),
                "scene_build_s": round(t_build, 2), "frames_in_flight": nbuf * batch, "launches_in_flight": nbuf,
                "frames_per_launch": batch, "use_beam": args.beam,
                "descriptor_table": (ctx.derived_info() if (args.pipeline == 1 and hasattr(ctx, "derived_info")) else None),
                "verification": vinfo,
            },
            "roofline": roof,
        }
        # ---- CPU baseline: the oracle on a bounded subsample of the same frames, 1 thread ----
        if args.cpu_seconds > 0 and ngpu == 1:
            from oracle import oracle
            okw = dict(bounces=args.bounces, mirror_mask=args.mirror, spp=args.spp, want_hits=False)
            t1 = time.perf_counter()
            probe = oracle.render(pool, W, H, cam, 2, args.mode, xstep=64, ystep=64, **okw)
            dt = max(time.perf_counter() - t1, 1e-4)
            per_px = dt / max(probe["stats"]["pixels"], 1)
            want_px = args.cpu_seconds / per_px
            stepxy = max(1, int(np.ceil(np.sqrt(W * H / want_px))))
            crays, cpix, nframes, dt = 0, 0, 0, 0.0
            t1 = time.perf_counter()
            while dt < args.cpu_seconds * 0.8 and nframes < 64:
                smp = oracle.render(pool, W, H, cam, 2 + nframes, args.mode, xstep=stepxy, ystep=stepxy, **okw)
                crays += smp["stats"]["rays"]
                cpix += smp["stats"]["pixels"]
                nframes += 1
                dt = time.perf_counter() - t1
            # informational, labelled separately (SURVEY 8d ii): the same oracle on all host cores, a few seconds
            ncores = usable_cores()
            if ncores > 1:
                t1 = time.perf_counter()
                arays, apass = 0, 0
                while time.perf_counter() - t1 < 4.0 and apass < 64:
                    smp = oracle.render(pool, W, H, cam, 2 + apass, args.mode, threads=ncores, **okw)
                    arays += smp["stats"]["rays"]
                    apass += 1
                adt = time.perf_counter() - t1
                line["cpu_all_cores"] = {"value": round(arays / adt / 1e6, 2), "unit": "Mrays/s", "cores": ncores, "kind": "port",
                                         "sample": "%d full frame(s), OpenMP over rows on the cores this process may use (affinity / cgroup "
                                                   "quota; os.cpu_count() = %d), %.1f s" % (apass, os.cpu_count() or 0, adt)}
            line["cpu_baseline"] = {
                "value": round(crays / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
                "sample": "every %d-th pixel in x and y of the same frames, %d pass(es) with frameNumber 2.. "
                          "(%d pixels, %d rays, %.1f s), single-threaded C oracle; this process may use %d of the host's %d cores" % (
                              stepxy, nframes, cpix, crays, dt, usable_cores(), os.cpu_count() or 0),
            }
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
        if args.verify and not verified:
            print("bench.py: verification against the oracle FAILED: " + str(vinfo), file=sys.stderr)
            rc = 1
    if dist.is_initialized():
        dist.barrier()
        if not stub:
            dist.destroy_process_group()
    ctx.close()
    if stub:
        return rc, (line if rank == 0 else None)
    sys.exit(rc)


if __name__ == "__main__":
    main()
