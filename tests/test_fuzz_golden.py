"""Pools no builder produces, under the reference shader itself (tests/golden/fuzz_golden.npz, made by
tests/golden/make_golden_fuzz.py on Mesa llvmpipe): random tag / value / normal mixes, child pointers that point
backwards, into the middle of other records and into the zero bytes behind the tree; every render mode, incl. the
ones trace() has no branch for; hostile cameras; frame numbers up to +-2^31; and the path options the shader carries
but does not run (more or fewer path segments, the mirror material), switched on in memory by the harness.  The oracle and the three HIP pipelines against those vectors, bit for bit."""
import os

import numpy as np
import pytest

from helpers import compare_with_golden
import helpers

HERE = os.path.dirname(os.path.abspath(__file__))
_Z = None


def _z():
    global _Z
    if _Z is None:
        _Z = np.load(os.path.join(HERE, "golden", "fuzz_golden.npz"))
    return _Z


def _cases():
    z = np.load(os.path.join(HERE, "golden", "fuzz_golden.npz"))
    return [tuple(s.split(":")) for s in z["index"]]


def _golden(name):
    z = _z()
    w, h, frame, mode, same = (int(v) for v in z[name + "/meta"])
    path = z[name + "/path"] if name + "/path" in z.files else (2, 0)
    return dict(w=w, h=h, frame=frame, mode=mode, patched_same=bool(same), cam=z[name + "/cam"], rgba=z[name + "/rgba"],
                depth_bits=z[name + "/depth_bits"], first_hit=z[name + "/first_hit"], bounces=int(path[0]), mirror_mask=int(path[1]))


def test_fixture_is_what_the_generator_describes():
    cases = _cases()
    assert len(cases) >= 390
    assert sum(1 for n, _ in cases if n.startswith("pv_")) == 40
    assert {pk[0] for _, pk in cases} == {"f", "m", "s", "t"}   # fuzz, mangled, scene, tiny
    assert {int(_z()[n + "/meta"][2]) for n, _ in cases} >= {2147483647, -2147483648, 16777216}
    assert {int(_z()[n + "/meta"][3]) for n, _ in cases} >= {0, 1, 2, 3, 4, 5, -1}


@pytest.mark.parametrize("name,poolkey", _cases())
def test_oracle_matches_reference_shader_on_fuzz_pools(name, poolkey):
    from oracle import oracle
    g = _golden(name)
    assert g["patched_same"]
    res = oracle.render(_z()["pool/" + poolkey], g["w"], g["h"], g["cam"], g["frame"], g["mode"], bounces=g["bounces"],
                        mirror_mask=g["mirror_mask"])
    bad = compare_with_golden(res, g)
    assert bad == {k: 0 for k in bad}, bad


@pytest.mark.gpu
@pytest.mark.parametrize("poolkey", sorted({pk for _, pk in _cases()}))
def test_hip_matches_reference_shader_on_fuzz_pools(poolkey):
    from svo_raytracer_amd import hiplib
    ctx = helpers.DualContext()
    try:
        pool = _z()["pool/" + poolkey]
        names = [n for n, pk in _cases() if pk == poolkey]
        for pipeline in (0, 1, 2):
            ctx.set_pipeline(pipeline)
            for i, name in enumerate(names):
                g = _golden(name)
                res = ctx.render(pool if i == 0 else None, g["w"], g["h"], g["cam"], g["frame"], g["mode"],
                                 bounces=g["bounces"], mirror_mask=g["mirror_mask"])
                bad = compare_with_golden(res, g)
                assert bad == {k: 0 for k in bad}, (name, pipeline, bad)
    finally:
        ctx.close()


def test_oracle_sin_cos_at_large_arguments_are_llvmpipe_s():
    """sin / cos of arguments up to 1.7e10 (frameNumber * 7.8 at frameNumber = +-2^31): the clamp of the result to [-1, 1]
    and the x86 float -> int conversion inside the range reduction, recorded from llvmpipe through a probe shader."""
    import ctypes
    from oracle import oracle
    L = oracle.lib()
    for fn in (L.svo_oracle_sin, L.svo_oracle_cos):
        fn.restype = ctypes.c_float
        fn.argtypes = [ctypes.c_float]
    z = _z()
    n = 0
    for key in z["sinprobe/index"]:
        x, ref = z["sinprobe/%s/x" % key], z["sinprobe/%s/ref_bits" % key]
        fn = L.svo_oracle_cos if key.endswith("_1") else L.svo_oracle_sin
        mine = np.array([fn(float(v)) for v in x], dtype=np.float32).view(np.uint32)
        assert np.array_equal(mine, ref), key
        n += int((np.abs(ref.view(np.float32)) == 1.0).sum())
    assert n > 1000   # the clamp is really exercised


def test_oracle_acos_and_fog_exp_are_llvmpipe_s():
    """acos() around +-1, 0, 0.5, beyond 1, NaN / inf; the fog term exp(-0.5 x 2) = exp2(x * -log2(e)) across its clamps
    (the smallest results are built in the exponent field: 0.0, not a denormal), NaN in -> NaN out.  45 000 arguments."""
    import ctypes
    from oracle import oracle
    L = oracle.lib()
    for fn in (L.svo_oracle_acos, L.svo_oracle_exp2):
        fn.restype = ctypes.c_float
        fn.argtypes = [ctypes.c_float]
    z = _z()
    with np.errstate(all="ignore"):
        for key in z["fnprobe/index"]:
            x = z["fnprobe/%s/x_bits" % key].view(np.float32)
            ref = z["fnprobe/%s/ref_bits" % key]
            if key.startswith("0_"):
                mine = np.array([L.svo_oracle_acos(float(v)) for v in x], dtype=np.float32)
            else:
                k = np.float32(-0.5 * 2 * 1.44269504)
                mine = np.array([L.svo_oracle_exp2(float(np.float32(v) * k)) for v in x], dtype=np.float32)
            both_nan = np.isnan(mine) & np.isnan(ref.view(np.float32))
            assert not ((mine.view(np.uint32) != ref) & ~both_nan).any(), key
