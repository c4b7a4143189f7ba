"""Fuzz parity: random valid pools (tests/fuzzpool.py), every render mode, both shallow and deep-embedded."""
import numpy as np
import pytest

import fuzzpool
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd.cameras import CAMERAS, rot_cam

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
    ctx = hiplib.HipContext(0)
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
