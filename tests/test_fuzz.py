"""Fuzz parity: random valid pools (tests/fuzzpool.py), every render mode, both shallow and deep-embedded."""
import numpy as np
import pytest

import fuzzpool
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd.cameras import CAMERAS, rot_cam
import helpers

CAMS = [CAMERAS["K0"], CAMERAS["K1"], rot_cam((1.45, 1.7, 2.9), -0.35, 0.1), rot_cam((1.5, 1.5, 1.5), 0.9, 2.0)]


def test_fuzz_pools_are_valid():
    for seed in range(4):
        pool = fuzzpool.random_pool(seed, p_interior=0.8, p_empty=0.8)
        rc, _, depth = scene.validate_pool(pool)
        assert rc == 0 and depth <= 6


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_fuzz_hip_matches_oracle(seed):
    from svo_raytracer_amd import hiplib
    from oracle import oracle
    pool = fuzzpool.random_pool(seed, max_depth=5 + seed % 2, p_interior=0.8, p_empty=0.8)
    if seed % 3 == 2:
        pool = scene.embed_deep(pool, 7)       # depth 12-13: LOD cap on bounce rays, Phong branch
    ctx = helpers.DualContext()
    try:
        for pipeline in (0, 1, 2):
            ctx.set_pipeline(pipeline)
            for ci, cam in enumerate(CAMS):
                cam = cam.copy()
                if seed % 3 == 2:
                    cam[:3] = (1.0 + (cam[:3].astype(np.float64) - 1.0) / 128).astype(np.float32)
                for mode in (0, 1, 2, 3):
                    got = ctx.render(pool, 88, 56, cam, 2 + seed, mode, bounces=2 + (seed % 2))
                    ref = oracle.render(pool, 88, 56, cam, 2 + seed, mode, bounces=2 + (seed % 2))
                    tag = (seed, pipeline, ci, mode)
                    assert (got["rgba"] == ref["rgba"]).all(), tag
                    assert (got["depth"].view(np.uint32) == ref["depth"].view(np.uint32)).all(), tag
                    for k in ("pointer", "value", "raw_normal", "level", "iter"):
                        assert (got["hits"][k] == ref["hits"][k]).all(), tag + (k,)
    finally:
        ctx.close()


def _mangled(pool, seed):
    """Bytes the builder never produces but a buggy or hostile host could upload: child pointers that point backwards
    (cycles), far past the end (reads give 0) or into the middle of other records; a truncated tail."""
    rng = np.random.RandomState(1000 + seed)
    p = pool.copy()
    interior = []
    stack = [0]
    while stack and len(interior) < 4000:       # offsets of reachable interior records
        o = stack.pop()
        cp = int.from_bytes(bytes(p[o + 1:o + 5]), "big", signed=True)
        if cp == 0:
            continue
        interior.append(o)
        mask = (int(p[o + 5]) << 8) | int(p[o + 6])
        c = o + cp
        for n in range(8):
            tag = (mask >> (2 * n)) & 3
            if tag == 0 and c + 7 <= p.size:
                stack.append(c)
            c += {0: 7, 1: 3, 2: 7, 3: 1}[tag]
    for o in rng.choice(interior[1:], size=min(12, len(interior) - 1), replace=False):
        kind = rng.randint(0, 3)
        if kind == 0:
            rel = -int(o)                                   # back to the root block: a cycle
        elif kind == 1:
            rel = int(p.size) + int(rng.randint(0, 1 << 20)) - int(o)   # past the end
        else:
            rel = int(rng.randint(7, p.size)) - int(o)     # somewhere inside the pool, any alignment
        p[o + 1:o + 5] = np.frombuffer(int(rel).to_bytes(4, "big", signed=True), dtype=np.uint8)
    return p[: p.size - int(rng.randint(0, 9))]


def test_oracle_beam_is_exact_on_fuzz_and_mangled_pools():
    """use_beam = 1 keeps pointer / value / normal / level / t and the images, whatever the bytes are (CPU statement)."""
    from oracle import oracle
    for seed in range(4):
        pool = fuzzpool.random_pool(seed, max_depth=5, p_interior=0.8, p_empty=0.8)
        for pl in (pool, _mangled(pool, seed)):
            for cam in (CAMS[0], CAMS[3]):
                for mode in (0, 2):
                    a = oracle.render(pl, 64, 40, cam, 3, mode)
                    b = oracle.render(pl, 64, 40, cam, 3, mode, use_beam=True)
                    assert (a["rgba"] == b["rgba"]).all() and (a["depth"].view(np.uint32) == b["depth"].view(np.uint32)).all()
                    for k in ("pointer", "value", "raw_normal", "level"):
                        assert (a["hits"][k] == b["hits"][k]).all(), (seed, mode, k)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_fuzz_mangled_pools_and_beam_match_oracle(seed):
    """Pools with cycles, out-of-range and misaligned child pointers, with and without the beam pre-pass: the three
    pipelines against the oracle (and no hang, no fault: every read is range-checked, every walk is bounded)."""
    from svo_raytracer_amd import hiplib
    from oracle import oracle
    base = fuzzpool.random_pool(seed, max_depth=5, p_interior=0.8, p_empty=0.8)
    ctx = helpers.DualContext()
    try:
        for pool in (base, _mangled(base, seed)):
            for pipeline in (0, 1, 2):
                ctx.set_pipeline(pipeline)
                for cam in (CAMS[0], CAMS[1], CAMS[3]):
                    for mode in (0, 2):
                        for beam in (0, 1):
                            got = ctx.render(pool, 80, 48, cam, 3, mode, use_beam=beam)
                            ref = oracle.render(pool, 80, 48, cam, 3, mode, use_beam=bool(beam))
                            tag = (seed, pipeline, mode, beam)
                            assert (got["rgba"] == ref["rgba"]).all(), tag
                            assert (got["depth"].view(np.uint32) == ref["depth"].view(np.uint32)).all(), tag
                            for k in ("pointer", "value", "raw_normal", "level", "iter"):
                                assert (got["hits"][k] == ref["hits"][k]).all(), tag + (k,)
                            if beam:
                                assert np.array_equal(ctx.read_beam().view(np.uint32), ref["beam"].view(np.uint32)), tag
    finally:
        ctx.close()
