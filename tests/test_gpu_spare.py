"""The spare-ray kernel (svo-raytracer_amd/csrc/variants/svo_persist2.hip.h + svo_travloop3.h; round 5's experiment on the lanes that
wait for a round, opt-in through SVO_SPARE=1 because it lost the throughput A/B: profiles/round5_experiments.txt) must leave the
bytes persist_kernel leaves -- every render mode, path options, samples and progressive sequences in one launch, frames with
their own cameras, the beam pre-pass -- and the reference shader's own image of the benchmark's configuration."""
import os

import numpy as np
import pytest

import poolcache

pytestmark = pytest.mark.gpu


def _frames(ctx, pool, cams):
    """a fixed programme of frames on a context; returns their images"""
    out = []
    w, h = 200, 120
    ctx.set_pipeline(1)
    ctx.pool_upload(pool)
    for mode, kw in ((0, {}), (1, {}), (2, {}), (3, {}), (4, {}), (0, dict(bounces=5, mirror_mask=0b1000)), (0, dict(spp=3)),
                     (0, dict(use_beam=1)), (2, dict(use_beam=1))):
        for cam in ("K1", "K2"):
            out.append(ctx.render(None, w, h, cams[cam], 7, mode, **kw))
    info = ctx.launch_info()
    # a progressive sequence in one launch, and frames with their own cameras in one launch, on a ring
    ctx.resize(w, h)
    ctx.set_camera(cams["K1"])
    ctx.set_params(2, 0, 0, 0, 2, 0, 1)
    ctx.set_progressive(True)
    ctx.set_sequence(5, True)
    ctx.ring_create(2, 1, want_hits=True)
    out.append(ctx.ring_read(ctx.ring_submit(2, 1), 0, want_hits=True))
    ctx.set_progressive(False)
    ctx.set_sequence(1, False)
    ctx.ring_create(2, 3, want_hits=True)
    s = ctx.ring_submit_cams(np.stack([cams["K1"], cams["K2"], cams["K0"]]), [1, 9, 4])
    out += [ctx.ring_read(s, k, want_hits=True) for k in range(3)]
    ctx.ring_destroy()
    return out, info


def test_spare_kernel_leaves_the_bytes_of_persist_kernel(monkeypatch):
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(256)
    res = {}
    for spare in ("1", "0"):
        monkeypatch.setenv("SVO_SPARE", spare)     # read by a context of libsvohip_variants.so when it sets up its persistent pipeline
        # "1": the spare-ray kernel, which only the variants library carries; "0": the library that ships (it reads no such switch)
        ctx = hiplib.HipContext(0, lib_path=hiplib.VARIANTS_LIB_PATH if spare == "1" else None)
        try:
            res[spare] = _frames(ctx, pool, CAMERAS)
        finally:
            ctx.close()
    # the two kernels differ in occupancy (their launch shape tells them apart: 4 against 6 waves per SIMD)
    assert res["1"][1]["waves_per_cu"] == 16 and res["0"][1]["waves_per_cu"] == 24, (res["1"][1], res["0"][1])
    assert len(res["1"][0]) == len(res["0"][0]) == 22
    for a, b in zip(res["1"][0], res["0"][0]):
        assert np.array_equal(a["rgba"], b["rgba"])
        assert np.array_equal(a["depth"].view(np.uint32), b["depth"].view(np.uint32))
        assert a["hits"].tobytes() == b["hits"].tobytes()


def test_spare_kernel_renders_the_reference_shader_s_config3_and_config4_frames(monkeypatch):
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from test_config3 import GOLD, _check, _meta
    z = np.load(GOLD)
    step = int(z["step"][0])
    pool = poolcache.pool()
    monkeypatch.setenv("SVO_SPARE", "1")
    ctx = hiplib.HipContext(0, lib_path=hiplib.VARIANTS_LIB_PATH)
    try:
        ctx.set_pipeline(1)
        ctx.pool_upload(pool)
        for name in ("c3_f2", "c3_K0_f3", "c3_m2", "c4_f2"):
            w, h, frame, mode, bounces, mirror = _meta(z, name)
            _check(ctx.render(None, w, h, z[name + "/cam"], frame, mode, bounces=bounces, mirror_mask=mirror), z, name, step)
        assert ctx.launch_info()["waves_per_cu"] == 16
        # the benchmark's shape: 3 submissions in flight x 4 frames, frame 2 read back out of its slot
        w, h, frame, mode, bounces, mirror = _meta(z, "c3_f2")
        ctx.resize(w, h)
        ctx.set_camera(z["c3_f2/cam"])
        ctx.set_params(2, mode, 0, 0, bounces, mirror, 1)
        ctx.ring_create(3, 4, want_hits=True)
        slots = [ctx.ring_submit(2 + 4 * b, 4) for b in range(3)]
        _check(ctx.ring_read(slots[0], 0, want_hits=True), z, "c3_f2", step)
        ctx.ring_destroy()
    finally:
        ctx.close()


def test_rounds_with_and_without_the_tables_leave_the_same_bytes(monkeypatch):
    """Round 5's tables (row / column tables per launch, the context's normal table: DESIGN.md section 4) hold values the round
    would compute with the same device functions; SVO_RC_TABLE=0 / SVO_NORMAL_TABLE=0 switch back to computing in the round.
    Same frames either way -- every render mode, path options, the beam pre-pass, frames with their own cameras, odd image sizes
    (rows of the tables that are not 16-byte multiples apart)."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(256)

    def programme(ctx):
        out = []
        ctx.set_pipeline(1)
        ctx.pool_upload(pool)
        for (w, h) in ((200, 120), (201, 67)):
            for mode, kw in ((0, {}), (1, {}), (2, {}), (3, {}), (0, dict(bounces=4, mirror_mask=0b1000)), (0, dict(use_beam=1))):
                out.append(ctx.render(None, w, h, CAMERAS["K1"], 9, mode, **kw))
        ctx.resize(201, 67)
        ctx.set_params(2, 0, 0, 0, 2, 0, 1)
        ctx.ring_create(2, 3, want_hits=True)
        s = ctx.ring_submit_cams(np.stack([CAMERAS["K1"], CAMERAS["K2"], CAMERAS["K0"]]), [1, 1 << 24, -7])
        out += [ctx.ring_read(s, k, want_hits=True) for k in range(3)]
        s = ctx.ring_submit(40, 3)
        out += [ctx.ring_read(s, k, want_hits=True) for k in range(3)]
        ctx.ring_destroy()
        return out

    res = {}
    for key, env in (("tables", {}), ("rounds", {"SVO_RC_TABLE": "0", "SVO_NORMAL_TABLE": "0"})):
        for k in ("SVO_RC_TABLE", "SVO_NORMAL_TABLE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        # "tables": the library that ships; "rounds": the variants library, which reads the two switches
        ctx = hiplib.HipContext(0, lib_path=hiplib.VARIANTS_LIB_PATH if env else None)
        try:
            res[key] = programme(ctx)
        finally:
            ctx.close()
    assert len(res["tables"]) == len(res["rounds"]) == 18
    for a, b in zip(res["tables"], res["rounds"]):
        assert np.array_equal(a["rgba"], b["rgba"])
        assert np.array_equal(a["depth"].view(np.uint32), b["depth"].view(np.uint32))
        assert a["hits"].tobytes() == b["hits"].tobytes()
