"""BASELINE.json config 1: 512^3 procedural SVO, 256x256 frame, primary rays only -- bit-exact hit-ID check
against the reference shader's own output (llvmpipe golden, tests/golden/make_golden_config1.py).
CPU leg: the oracle (stands in for the 'Java CPU traversal' of that config).  GPU leg: the HIP path."""
import os
import zlib

import numpy as np
import pytest

import svo_raytracer_amd.scene as scene
import helpers

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config1_512.npz")
CASES = [("K0", 1), ("K0", 3), ("K1", 1)]


@pytest.fixture(scope="module")
def pool512():
    z = np.load(GOLD)
    pool, _ = scene.build_scene(512)
    assert pool.size == int(z["pool_size"][0]) and zlib.crc32(pool.tobytes()) == int(z["pool_crc32"][0]), \
        "scene generator drifted: regenerate tests/golden/config1_512.npz"
    return pool


def _check(res, z, key):
    fh = z[key + "/first_hit"]
    hit = fh[..., 0] != 0
    assert (res["rgba"] == z[key + "/rgba"]).all()
    assert (res["depth"].view(np.uint32) == z[key + "/depth_bits"]).all()
    h = res["hits"]
    assert (h["pointer"] == fh[..., 0]).all()                 # hit voxel IDs
    assert ((h["value"] == fh[..., 1]) | ~hit).all()
    assert ((h["raw_normal"] == fh[..., 2]) | ~hit).all()     # packed normals
    assert ((h["level"] == (fh[..., 3] >> 16)) | ~hit).all()
    assert ((h["iter"] == (fh[..., 3] & 0xFFFF)) | ~hit).all()


@pytest.mark.parametrize("camname,mode", CASES)
def test_config1_cpu_traversal_matches_reference(pool512, camname, mode):
    from oracle import oracle
    z = np.load(GOLD)
    key = "%s_m%d" % (camname, mode)
    _check(oracle.render(pool512, 256, 256, z[key + "/cam"], 2, mode), z, key)


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1, 2])
@pytest.mark.parametrize("camname,mode", CASES)
def test_config1_hip_matches_reference(pool512, camname, mode, pipeline):
    from svo_raytracer_amd import hiplib
    z = np.load(GOLD)
    key = "%s_m%d" % (camname, mode)
    ctx = helpers.DualContext()
    try:
        ctx.set_pipeline(pipeline)
        _check(ctx.render(pool512, 256, 256, z[key + "/cam"], 2, mode), z, key)
    finally:
        ctx.close()
