"""incremental pool updates from SDF brush edits (Main.placeSDF, Main.java:338-353:
Octree.useSDFBrush produces two byte ranges, Renderer.updateSSBO sends them).  The product side of that row is
svo_pool_update; the edits themselves come from the oracle-side restatement of Octree.useSDFBrush /
subdivideNode / ChangeBounds (oracle/octree_restatement.cpp, Octree.java:672-885; parity unpinned: no JDK)."""
import numpy as np
import pytest

from svo_raytracer_amd import hostlib
from oracle import octree as restated
import svo_raytracer_amd.scene as scene
from svo_raytracer_amd.cameras import CAMERAS, rot_cam
import helpers

EDITS = [
    ("sphere", (20, 24, 40), 7, 2),     # add
    ("sphere", (44, 19, 24), 6, 0),     # carve
    ("sphere", (30, 21, 30), 9, 1),     # big fill: interior nodes re-tagged 2, children marked 127
]
KEDIT = rot_cam((1.5, 1.62, 1.55), -1.15, 0.4)


def _walk(pool):
    """(offset, tag, record bytes) of every reachable node"""
    out = []
    stack = [0]
    while stack:
        p = stack.pop()
        cp = int.from_bytes(bytes(pool[p + 1:p + 5]), "big", signed=True)
        mask = (int(pool[p + 5]) << 8) | int(pool[p + 6])
        if cp == 0:
            continue
        c = p + cp
        for n in range(8):
            tag = (mask >> (2 * n)) & 3
            sz = {0: 7, 1: 3, 2: 7, 3: 1}[tag]
            out.append((c, tag, bytes(pool[c:c + sz])))
            if tag == 0:
                stack.append(c)
            c += sz
    return out


def test_brush_edits_stay_inside_change_bounds_and_keep_the_pool_valid():
    pool, _ = scene.build_scene(64)
    o = hostlib.Octree(4096)
    o.adopt(pool)
    for kind, org, r, val in EDITS:
        before = o.getByteBuffer()
        cb = restated.useSDFBrushSphere(o, org, r, val, worldSize=64, maxLOD=6)
        after = o.getByteBuffer()
        n = before.size
        diff = np.nonzero(before != after[:n])[0]
        inside = ((diff >= cb[0]) & (diff < cb[1])) | ((diff >= cb[2]) & (diff < cb[3]))
        assert inside.all(), "a modified byte lies outside both update ranges"
        assert cb[2] == n and cb[3] == after.size          # new nodes are appended: [start1, end1) = the tail
        rc, _, depth = scene.validate_pool(after)
        assert rc == 0 and depth <= 6
    nodes = _walk(o.getByteBuffer())
    # quirk Q5: a filled former interior node is re-tagged 2 but keeps its old child pointer / mask bytes
    assert any(tag == 2 and any(rec[1:]) for _, tag, rec in nodes)
    # children of a filled node are only marked with DELETE_VALUE, not reclaimed (Constants.java:16)
    pool2 = o.getByteBuffer()
    assert (pool2 == 127).any()


def test_box_brush_matches_reference_distance_rule():
    o = hostlib.Octree(4096)
    pool, _ = scene.build_scene(64)
    o.adopt(pool)
    cb = restated.useSDFBrushBox(o, (34, 30, 30), 3, 5, 4, 3, worldSize=64, maxLOD=6)
    assert cb[3] > cb[2]
    assert scene.validate_pool(o.getByteBuffer())[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_ranged_updates_render_like_a_full_upload(pipeline):
    """Main.placeSDF sends two ranges (Main.java:349-350); the GPU pool must then render exactly like the
    edited pool uploaded whole, and like the oracle."""
    from svo_raytracer_amd import hiplib
    from oracle import oracle
    pool, _ = scene.build_scene(64)
    o = hostlib.Octree(4096)
    o.adopt(pool)
    ctx = helpers.DualContext()
    try:
        ctx.set_pipeline(pipeline)
        ctx.pool_upload(pool)
        for kind, org, r, val in EDITS:
            cb = restated.useSDFBrushSphere(o, org, r, val, worldSize=64, maxLOD=6)
            host = o.getByteBuffer()
            if cb[1] > cb[0]:
                ctx.pool_update(host, cb[0], cb[1])
            if cb[3] > cb[2]:
                ctx.pool_update(host, cb[2], cb[3])
            for mode in (0, 2):
                got = ctx.render(None, 96, 64, KEDIT, 2, mode)
                ref = oracle.render(host, 96, 64, KEDIT, 2, mode)
                assert (got["rgba"] == ref["rgba"]).all()
                assert (got["depth"].view(np.uint32) == ref["depth"].view(np.uint32)).all()
                for k in ("pointer", "value", "raw_normal", "level", "iter"):
                    assert (got["hits"][k] == ref["hits"][k]).all(), k
            assert (ctx.pool_download(host.size) == host).all()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_descriptor_table_follows_brush_strokes_without_a_rebuild():
    """svo_pool_update keeps a walkable descriptor table up to date (derive::refresh_table: states whose child block a
    range touches are recomputed, changed sibling groups and new subtrees appended) instead of dropping it: a sequence
    of brush strokes -- fills that re-tag interior nodes, carves, strokes over earlier strokes -- must render exactly like
    the oracle after every stroke, on the table, and the table must not have been rebuilt in between."""
    from svo_raytracer_amd import hiplib
    from oracle import oracle
    n, lod = 128, 7
    pool, _ = scene.build_scene(n)
    o = hostlib.Octree(1 << 16)
    o.adopt(pool)
    rng = np.random.default_rng(11)
    cams = [KEDIT, CAMERAS["K1"]]
    ctx = helpers.DualContext()
    try:
        ctx.set_pipeline(1)
        ctx.set_derived(1)
        ctx.pool_upload(pool)
        info = ctx.derived_info()
        assert info["walkable"]
        followed = issued = 0
        prev_host = pool
        for stroke in range(14):
            org = tuple(int(v) for v in rng.integers(n // 4, 3 * n // 4, 3))
            r = int(rng.integers(3, 14))
            val = int(rng.choice([0, 0, 1, 2, 3]))
            if stroke % 5 == 4:
                cb = restated.useSDFBrushBox(o, org, r, max(2, r // 2), r, val, worldSize=n, maxLOD=lod)
            else:
                cb = restated.useSDFBrushSphere(o, org, r, val, worldSize=n, maxLOD=lod)
            host = o.getByteBuffer()
            before = ctx.derived_refresh_info()["refreshes"]
            # (a stroke that rewrites the root record -- its mask changes when one of the eight top-level children is
            # re-tagged, common in a 128^3 world, out of reach of a brush at 8192^3 -- is the one case left to a rebuild)
            root_changed = bool((host[1:7] != prev_host[1:7]).any())
            n_updates = 0
            if cb[1] > cb[0]:
                ctx.pool_update(host, cb[0], cb[1]); n_updates += 1
            if cb[3] > cb[2]:
                ctx.pool_update(host, cb[2], cb[3]); n_updates += 1
            prev_host = host
            after = ctx.derived_refresh_info()
            followed += after["refreshes"] - before
            if not root_changed:
                issued += n_updates
                assert after["refreshes"] - before == n_updates, (stroke, cb)
            for cam in cams:
                got = ctx.render(None, 128, 80, cam, 2 + stroke, 0)
                ref = oracle.render(host, 128, 80, cam, 2 + stroke, 0)
                assert (got["rgba"] == ref["rgba"]).all(), stroke
                assert (got["depth"].view(np.uint32) == ref["depth"].view(np.uint32)).all(), stroke
                for k in ("pointer", "value", "raw_normal", "level", "iter"):
                    assert (got["hits"][k] == ref["hits"][k]).all(), (stroke, k)
            assert ctx.derived_info()["walkable"]
        assert issued >= 10 and followed >= issued, (followed, issued)   # none of those fell back to a rebuild
        # the same pool uploaded whole into a fresh table renders the same bytes (and has no garbage groups)
        grown = ctx.derived_info()["descriptors"]
        ctx.pool_upload(o.getByteBuffer())
        fresh = ctx.derived_info()["descriptors"]
        assert fresh <= grown
    finally:
        ctx.close()
