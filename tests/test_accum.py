"""The reference's dormant cross-frame accumulation (svotrace.comp:712-719, commented out; MAX_FRAME_ITER :43) exposed as
svo_set_progressive.  Pinned by the reference shader itself: tests/golden/accum_golden.npz holds sequences of consecutive
frames rendered under llvmpipe with that block switched on in memory (tests/golden/make_golden_accum.py), each frame
blending into the image the previous one left."""
import os

import numpy as np
import pytest
import helpers

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "accum_golden.npz")
PIPELINES = [int(v) for v in os.environ.get("SVO_TEST_PIPELINES", "0,1,2").split(",")]


def sequences():
    z = np.load(GOLDEN)
    out = []
    for name in z["index"].tolist():
        meta = z[name + "/meta"]
        out.append({"name": name, "n": int(meta[0]), "w": int(meta[1]), "h": int(meta[2]), "mode": int(meta[3]),
                    "frames": [int(v) for v in meta[4:]], "cam": z[name + "/cam"], "rgba": z[name + "/rgba"],
                    "depth_bits": z[name + "/depth_bits"]})
    return out


@pytest.mark.parametrize("seq", sequences(), ids=lambda s: s["name"])
def test_oracle_accumulation_matches_the_reference_shader(seq):
    import svo_raytracer_amd.scene as scene
    from oracle import oracle
    pool, _ = scene.build_scene(seq["n"])
    last = np.zeros((seq["h"], seq["w"], 4), dtype=np.uint8)          # a fresh GL texture
    for i, f in enumerate(seq["frames"]):
        r = oracle.render(pool, seq["w"], seq["h"], seq["cam"], f, seq["mode"], last_rgba=last)
        assert (r["rgba"] == seq["rgba"][i]).all(), (seq["name"], f)
        assert (r["depth"].view(np.uint32) == seq["depth_bits"][i]).all()
        last = r["rgba"]
    if seq["frames"][-1] >= 100:                                        # frozen from MAX_FRAME_ITER on
        assert (seq["rgba"][-1] == seq["rgba"][-2]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", PIPELINES)
@pytest.mark.parametrize("seq", sequences(), ids=lambda s: s["name"])
def test_hip_accumulation_matches_the_reference_shader(seq, pipeline):
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    pool, _ = scene.build_scene(seq["n"])
    c = helpers.DualContext()              # fresh context = cleared images, like fresh GL textures
    try:
        c.set_pipeline(pipeline)
        c.set_progressive(True)
        c.pool_upload(pool)
        c.resize(seq["w"], seq["h"])
        c.set_camera(seq["cam"])
        for i, f in enumerate(seq["frames"]):
            c.set_params(f, seq["mode"], 0, 0, 2, 0, 1)
            c.dispatch()
            assert (c.read_color() == seq["rgba"][i]).all(), (seq["name"], f)
            assert (c.read_depth().view(np.uint32) == seq["depth_bits"][i]).all()
        c.set_progressive(False)          # and the live shader's behaviour is back
        c.set_params(seq["frames"][0], seq["mode"], 0, 0, 2, 0, 1)
        c.dispatch()
        plain = c.read_color()
        from oracle import oracle
        ref = oracle.render(pool, seq["w"], seq["h"], seq["cam"], seq["frames"][0], seq["mode"])
        assert (plain == ref["rgba"]).all()
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", PIPELINES)
def test_hip_accumulation_with_samples_stripes_and_beam_matches_oracle(pipeline):
    """Combinations the goldens do not hold (spp > 1 resolves through the sample accumulator first; packed stripes;
    the beam pre-pass): HIP against the oracle, frame after frame on one image."""
    import torch
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    from svo_raytracer_amd.tiles import stripe_layout, deinterleave
    from oracle import oracle
    pool, _ = scene.build_scene(256)
    w, h, world = 120, 84, 2
    c = helpers.DualContext()
    try:
        c.set_pipeline(pipeline)
        c.set_progressive(True)
        c.pool_upload(pool)
        c.resize(w, h)
        c.set_camera(CAMERAS["K1"])
        last = np.zeros((h, w, 4), dtype=np.uint8)
        for f in (2, 3, 4):
            c.set_params(f, 0, 0, 1, 2, 0, 3)       # use_beam = 1, spp = 3
            c.dispatch()
            ref = oracle.render(pool, w, h, CAMERAS["K1"], f, 0, spp=3, last_rgba=last)
            got = c.read_color()
            assert (got == ref["rgba"]).all(), f
            last = got
        # packed stripes into a caller-owned gather buffer that persists between frames
        rpr = stripe_layout(h, world, 0)[4]
        col = torch.zeros((rpr * world, w), dtype=torch.int32, device="cuda")
        dep = torch.zeros((rpr * world, w), dtype=torch.float32, device="cuda")
        c.bind_outputs(col.data_ptr(), dep.data_ptr(), None)
        last = np.zeros((h, w, 4), dtype=np.uint8)
        for f in (2, 3):
            c.set_params(f, 2, 0, 0, 2, 0, 1)
            for r in range(world):
                first, step, n, out0, rows = stripe_layout(h, world, r)
                c.set_stripes(first, step, n, out0)
                c.dispatch()
            torch.cuda.synchronize()
            got = deinterleave(col.cpu().numpy().view(np.uint8).reshape(rpr * world, w, 4), world, rpr, h)
            ref = oracle.render(pool, w, h, CAMERAS["K1"], f, 2, last_rgba=last)
            assert (got == ref["rgba"]).all(), f
            last = got
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", PIPELINES)
@pytest.mark.parametrize("seq", sequences(), ids=lambda s: s["name"])
def test_hip_sequences_in_one_dispatch_match_the_reference_shader(seq, pipeline):
    """svo_set_sequence: every prefix of a golden sequence as ONE dispatch (on the persistent pipeline one launch, the recurrence
    applied in frame order afterwards) = the shader's frame at the end of that prefix -- GI converging, the shadow mode, and the
    sequence that crosses MAX_FRAME_ITER (frames 98..101: the image freezes from frame 100 on); then the same sequence in two
    halves, the second continuing on the image the first left (fresh = 0)."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    pool, _ = scene.build_scene(seq["n"])
    c = helpers.DualContext()
    try:
        c.set_pipeline(pipeline)
        c.set_progressive(True)
        c.pool_upload(pool)
        c.resize(seq["w"], seq["h"])
        c.set_camera(seq["cam"])
        frames = seq["frames"]
        for n in range(1, len(frames) + 1):
            c.set_sequence(n, fresh=True)
            c.set_params(frames[0], seq["mode"], 0, 0, 2, 0, 1)
            c.dispatch()
            assert (c.read_color() == seq["rgba"][n - 1]).all(), (seq["name"], n)
            assert (c.read_depth().view(np.uint32) == seq["depth_bits"][n - 1]).all(), (seq["name"], n)
        half = len(frames) // 2
        if half >= 1:
            c.set_sequence(half, fresh=True)
            c.set_params(frames[0], seq["mode"], 0, 0, 2, 0, 1)
            c.dispatch()
            c.set_sequence(len(frames) - half, fresh=False)
            c.set_params(frames[half], seq["mode"], 0, 0, 2, 0, 1)
            c.dispatch()
            assert (c.read_color() == seq["rgba"][-1]).all(), seq["name"]
            assert (c.read_depth().view(np.uint32) == seq["depth_bits"][-1]).all(), seq["name"]
    finally:
        c.close()
