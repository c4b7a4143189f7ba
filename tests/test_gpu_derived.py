"""The interior-descriptor table (svo-raytracer_amd/csrc/svo_derive.hip.h) against the reference shader's goldens.

Pipeline 1 walks the table when the pool is derivable and the pool's records otherwise; every other GPU test runs it
in its default mode (table when possible).  Here: both modes on every golden pool -- the llvmpipe goldens, and the 397
fuzz / mangled cases (pools with overlapping, backward and out-of-range child pointers) -- bit for bit, which of them
are walkable, that builder-made pools always are, and that the table follows ranged pool updates."""
import os

import numpy as np
import pytest

from helpers import compare_with_golden, golden_case, golden_cases

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ctx():
    from svo_raytracer_amd import hiplib
    c = hiplib.HipContext(0)
    c.set_pipeline(1)
    yield c
    c.set_derived(1)
    c.close()


def _same(a, b):
    bad = {"rgba": int((a["rgba"] != b["rgba"]).any(axis=2).sum()),
           "depth": int((a["depth"].view(np.uint32) != b["depth"].view(np.uint32)).sum())}
    for k in ("pointer", "value", "raw_normal", "level", "iter"):
        bad[k] = int((a["hits"][k] != b["hits"][k]).sum())
    bad["t"] = int((a["hits"]["t"].view(np.uint32) != b["hits"]["t"].view(np.uint32)).sum())
    return bad


@pytest.mark.parametrize("derived", [1, 0])
@pytest.mark.parametrize("name,poolkey", golden_cases())
def test_both_walks_match_reference_shader_golden(ctx, name, poolkey, derived):
    g = golden_case(name, poolkey)
    ctx.set_derived(derived)
    res = ctx.render(g["pool"], g["w"], g["h"], g["cam"], g["frame"], g["mode"])
    bad = compare_with_golden(res, g)
    assert bad == {k: 0 for k in bad}, bad


def test_builder_made_golden_pools_are_walkable(ctx):
    """Every pool of the llvmpipe fixture that a builder made (terrains, embeddings, dust, brush-edited) must take the
    table path; the report says how many descriptors each needs."""
    from helpers import golden
    z = golden()
    ctx.set_derived(1)
    seen = {}
    for name, poolkey in golden_cases():
        if poolkey in seen:
            continue
        pool = z["pool/" + poolkey]
        ctx.pool_upload(pool)
        seen[poolkey] = (pool.size, ctx.derived_info())
    print({k: (n, i["descriptors"], i["walkable"]) for k, (n, i) in seen.items()})
    assert all(i["walkable"] for _, i in seen.values()), {k: i for k, (n, i) in seen.items() if not i["walkable"]}
    # a proper tree has at most one state per 15 bytes of pool (an interior record + the smallest child block); the
    # phantom state's children that are not records of the tree unroll into a few more
    assert all(i["descriptors"] <= n // 8 + 4096 for n, i in seen.values())


def _fuzz():
    return np.load(os.path.join(HERE, "golden", "fuzz_golden.npz"))


def _fuzz_cases():
    return [tuple(s.split(":")) for s in _fuzz()["index"]]


@pytest.mark.parametrize("poolkey", sorted({pk for _, pk in _fuzz_cases()}))
def test_both_walks_match_reference_shader_on_fuzz_pools(ctx, poolkey):
    z = _fuzz()
    pool = z["pool/" + poolkey]
    names = [n for n, pk in _fuzz_cases() if pk == poolkey]
    for derived in (1, 0):
        ctx.set_derived(derived)
        for i, name in enumerate(names):
            w, h, frame, mode, same = (int(v) for v in z[name + "/meta"])
            path = z[name + "/path"] if name + "/path" in z.files else (2, 0)
            g = dict(rgba=z[name + "/rgba"], depth_bits=z[name + "/depth_bits"], first_hit=z[name + "/first_hit"])
            res = ctx.render(pool if i == 0 else None, w, h, z[name + "/cam"], frame, mode, bounces=int(path[0]),
                             mirror_mask=int(path[1]))
            bad = compare_with_golden(res, g)
            assert bad == {k: 0 for k in bad}, (name, derived, bad)


def test_fuzz_pools_walkability_report(ctx):
    """Which of the fuzz fixture's pools the table can state (the others fall back to the records): the scene and tiny
    pools must be among them; mangled ones may or may not."""
    z = _fuzz()
    ctx.set_derived(1)
    out = {}
    for pk in sorted({pk for _, pk in _fuzz_cases()}):
        ctx.pool_upload(z["pool/" + pk])
        out[pk] = ctx.derived_info()
    walk = sorted(k for k, i in out.items() if i["walkable"])
    print("walkable:", len(walk), "of", len(out), "| not:", sorted(k for k in out if k not in walk))
    assert all(out[k]["walkable"] for k in out if k[0] in "st"), [k for k in out if k[0] in "st" and not out[k]["walkable"]]
    assert len(walk) >= len(out) // 2


@pytest.mark.parametrize("n,cam,mode", [(256, "K1", 0), (512, "K0", 2), (1024, "K2", 0)])
def test_table_walk_equals_record_walk_and_oracle(ctx, n, cam, mode):
    import svo_raytracer_amd.scene as scene
    from oracle import oracle
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(n)
    ctx.set_derived(1)
    a = ctx.render(pool, 320, 200, CAMERAS[cam], 3, mode)
    info = ctx.derived_info()
    assert info["walkable"] and 2 < info["descriptors"] <= pool.size // 15 + 4096
    ctx.set_derived(0)
    b = ctx.render(None, 320, 200, CAMERAS[cam], 3, mode)
    assert _same(a, b) == {k: 0 for k in _same(a, b)}
    ref = oracle.render(pool, 320, 200, CAMERAS[cam], 3, mode)
    bad = _same(a, ref)
    assert bad == {k: 0 for k in bad}, bad


def test_table_follows_ranged_pool_updates(ctx):
    """svo_pool_update changes records in place and appends new ones (Octree.useSDFBrush -> Renderer.updateSSBO,
    Main.java:349-350): the next dispatch must walk the edited tree."""
    import svo_raytracer_amd.scene as scene
    from oracle import oracle
    from svo_raytracer_amd.cameras import CAMERAS
    a, _ = scene.build_scene(128)
    b, _ = scene.build_scene(256)   # a different tree: stands in for an arbitrary edit, uploaded range by range
    ctx.set_derived(1)
    ctx.render(a, 160, 96, CAMERAS["K1"], 2, 0)
    d0 = ctx.derived_info()["descriptors"]
    big = np.zeros(max(a.size, b.size), np.uint8)
    big[:b.size] = b
    third = b.size // 3
    for s, e in ((0, third), (third, 2 * third), (2 * third, b.size)):
        ctx.pool_update(big, s, e)
    if a.size > b.size:
        ctx.pool_update(big, b.size, a.size)   # zero what is left of the old pool
    res = ctx.render(None, 160, 96, CAMERAS["K1"], 2, 0)
    assert ctx.derived_info()["descriptors"] != d0
    ref = oracle.render(b, 160, 96, CAMERAS["K1"], 2, 0)
    bad = _same(res, ref)
    assert bad == {k: 0 for k in bad}, bad


def _reachable_records(pool):
    """(offset, tag) of every record a proper walk from the root reaches"""
    out, stack = [], [0]
    while stack:
        p = stack.pop()
        cp = int.from_bytes(bytes(pool[p + 1:p + 5]), "big", signed=True)
        if cp == 0:
            continue
        mask = (int(pool[p + 5]) << 8) | int(pool[p + 6])
        c = p + cp
        for k in range(8):
            tag = (mask >> (2 * k)) & 3
            out.append((c, tag))
            if tag == 0:
                stack.append(c)
            c += {0: 7, 1: 3, 2: 7, 3: 1}[tag]
    return out


def test_refreshed_table_equals_a_rebuilt_one_under_random_record_edits(ctx):
    """derive::refresh_table against the full build: forty small edits of a kind no brush makes -- value bytes zeroed
    and set, child pointers cut and re-pointed at other blocks (shared and cyclic subtrees), tag masks rewritten,
    records copied over one another -- each sent through svo_pool_update.  After every edit the context whose table
    followed the update renders what a context that was given the whole pool anew renders, on the table and on the
    records, bit for bit (whether or not the edited pool is still walkable), and now and then what the oracle renders."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd import hiplib
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    pool, _ = scene.build_scene(64)
    host = pool.copy()
    recs = _reachable_records(host)
    inner = [o for o, t in recs if t == 0 and int.from_bytes(bytes(host[o + 1:o + 5]), "big") != 0]
    rng = np.random.default_rng(23)
    fresh = hiplib.HipContext(0)
    try:
        fresh.set_pipeline(1)
        ctx.set_derived(1)
        ctx.pool_upload(host)
        assert ctx.derived_info()["walkable"]
        followed = 0
        for edit in range(40):
            kind = edit % 5
            o, tag = recs[int(rng.integers(len(recs)))]
            if kind == 0:      # a value byte flips between empty and solid
                lo, hi = o, o + 1
                host[o] = 0 if host[o] else 3
            elif kind == 1:    # an interior node loses its children
                o = inner[int(rng.integers(len(inner)))]
                lo, hi = o + 1, o + 5
                host[lo:hi] = 0
            elif kind == 2:    # ... or gets another node's: a shared (possibly cyclic) subtree
                o, src = inner[int(rng.integers(len(inner)))], inner[int(rng.integers(len(inner)))]
                tgt = src + int.from_bytes(bytes(host[src + 1:src + 5]), "big", signed=True)
                lo, hi = o + 1, o + 7
                host[o + 1:o + 5] = np.frombuffer(int(tgt - o).to_bytes(4, "big", signed=True), np.uint8)
                host[o + 5:o + 7] = host[src + 5:src + 7]
            elif kind == 3:    # a tag mask is rewritten: the children's records are re-read at other sizes
                o = inner[int(rng.integers(len(inner)))]
                lo, hi = o + 5, o + 7
                host[lo:hi] = rng.integers(0, 256, 2, dtype=np.uint8)
            else:              # seven bytes of one record land on another
                src = recs[int(rng.integers(len(recs)))][0]
                lo, hi = o, min(o + 7, host.size)
                host[lo:hi] = host[src:src + (hi - lo)].copy()
            if lo < 7:
                continue       # (the root record: a rebuild by design, covered elsewhere)
            before = ctx.derived_refresh_info()["refreshes"]
            ctx.pool_update(host, lo, hi)
            followed += ctx.derived_refresh_info()["refreshes"] - before
            for derived in (1, 0):
                ctx.set_derived(derived)
                fresh.set_derived(derived)
                a = ctx.render(None, 128, 80, CAMERAS["K1"], 3 + edit, 0)
                b = fresh.render(host, 128, 80, CAMERAS["K1"], 3 + edit, 0)
                bad = _same(a, b)
                assert bad == {k: 0 for k in bad}, (edit, kind, derived, bad)
            ctx.set_derived(1)
            if edit % 8 == 7:
                ref = oracle.render(host, 128, 80, CAMERAS["K1"], 3 + edit, 0)
                bad = _same(a, ref)
                assert bad == {k: 0 for k in bad}, (edit, bad)
        assert followed >= 10, followed
    finally:
        fresh.close()


@pytest.mark.parametrize("waves", [1, 2])
def test_many_rays_per_lane_on_stale_stack_columns(ctx, waves):
    """A lane of a persistent wave casts ray after ray over the same LDS stack column.  trav_loop2's POP reads its level
    without a pushed-levels mask, so what a ray finds there must never depend on the rays before it: every ray starts on
    a zeroed column (DescWalk::fresh_stack; tests/test_kernel_isa.py checks the stores are in the shipped kernel).  Here
    the pressure test: one / two waves per CU over a full-HD frame -- 8 000 rays per lane, hostile cameras included --
    against the one-thread-per-pixel pipeline, whose stack is private to its one path."""
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    pool, _ = scene.build_scene(512)
    ctx.set_derived(1)
    cams = [CAMERAS["K1"], CAMERAS["K2"]]
    edge = np.array(CAMERAS["K0"], dtype=np.float32).copy()
    edge[:3] = (1.5, 1.25, 1.75)          # the camera on cell boundaries of three levels
    cams.append(edge)
    tilt = np.array(CAMERAS["K1"], dtype=np.float32).copy()
    tilt[3:] = np.float32(tilt[3:]) * np.float32([1, 0, 1] * 4) + np.float32([0, -1e-20, 0] * 4)   # axis-parallel rays: |d.y| < epsilon
    cams.append(tilt)
    for mode in (0, 2):
        for cam in cams:
            ctx.set_pipeline(0)
            ref = ctx.render(pool, 1920, 1080, cam, 3, mode)
            ctx.set_pipeline(1)
            ctx.set_tuning(waves, 9)
            got = ctx.render(None, 1920, 1080, cam, 3, mode)
            ctx.set_tuning(0, 0)
            bad = _same(got, ref)
            assert bad == {k: 0 for k in bad}, (mode, cam[:3], bad)
