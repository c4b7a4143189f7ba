"""The CPU oracle against the reference shader's own outputs (llvmpipe golden vectors).
Bit-exact bar: rgba8, depth bits, first-cast hit pointer / value / leafMask / level / iter."""
import numpy as np
import pytest

from helpers import compare_with_golden, golden_case, golden_cases
from oracle import oracle


@pytest.mark.parametrize("name,poolkey", golden_cases())
def test_oracle_matches_reference_shader(name, poolkey):
    g = golden_case(name, poolkey)
    assert g["patched_same"], "instrumented reference run diverged from the unmodified one"
    res = oracle.render(g["pool"], g["w"], g["h"], g["cam"], g["frame"], g["mode"])
    bad = compare_with_golden(res, g)
    assert bad == {k: 0 for k in bad}, bad


def test_oracle_pinned_transcendentals_are_deterministic():
    L = oracle.lib()
    # spot values recorded from the llvmpipe-pinned algorithms (regression guard for the oracle itself)
    assert L.svo_oracle_sin(0.0) == 0.0
    assert L.svo_oracle_cos(0.0) == 1.0
    assert abs(L.svo_oracle_acos(1.0)) < 1e-6
    assert L.svo_oracle_exp2(0.0) == 1.0
    r = L.svo_oracle_rand(3.0, 5.0)
    assert 0.0 <= r < 1.0


def test_oracle_on_all_cores_gives_the_same_bytes_and_counters():
    """svo_oracle_render_mt (the all-core CPU figure of bench.py): rows on OpenMP threads, identical output and stats."""
    import os
    import numpy as np
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    pool, _ = scene.build_scene(128)
    for mode in (0, 2):
        a = oracle.render(pool, 97, 61, CAMERAS["K1"], 3, mode, ystep=2)
        b = oracle.render(pool, 97, 61, CAMERAS["K1"], 3, mode, ystep=2, threads=max(2, os.cpu_count() or 2))
        assert (a["rgba"] == b["rgba"]).all() and (a["depth"].view(np.uint32) == b["depth"].view(np.uint32)).all()
        assert a["hits"].tobytes() == b["hits"].tobytes()
        for k in ("pixels", "rays", "nan_rays", "iterations", "alg_bytes", "max_iter"):
            assert a["stats"][k] == b["stats"][k], k
