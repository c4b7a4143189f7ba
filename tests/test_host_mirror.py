"""Host-side mirror of the reference's Java classes (Camera / Octree): CPU-only logic."""
import numpy as np
import pytest

from svo_raytracer_amd import hostlib
import svo_raytracer_amd.scene as scene


def test_camera_defaults_match_reference():
    c = hostlib.Camera()
    u = c.getUniform()
    # Camera.java:11-21
    assert np.allclose(u, [0, 0, 0, -1.6, -0.9, -1, -1.6, 0.9, -1, 1.6, -0.9, -1, 1.6, 0.9, -1])
    c.setPos(1.5, 1.5, 2.0)  # Main.java:120
    assert np.allclose(c.getUniform()[:3], [1.5, 1.5, 2.0])
    assert np.allclose(c.dir, [0, 0, 1])


def test_camera_rotate_is_a_rotation_and_accumulates():
    c = hostlib.Camera()
    base = c.getUniform()[3:].reshape(4, 3).astype(np.float64)
    c.rotate(0.0, 0.3, 0.0)
    c.rotate(-0.2, 0.0, 0.0)
    got = c.getUniform()[3:].reshape(4, 3).astype(np.float64)
    # lengths and pairwise angles are preserved
    assert np.allclose(np.linalg.norm(got, axis=1), np.linalg.norm(base, axis=1), atol=1e-5)
    assert np.allclose(got @ got.T, base @ base.T, atol=1e-4)
    # yaw 0.3 about +y then pitch about the camera's right axis: compare with float64 math
    def rot_axis(a, ax):
        ax = np.asarray(ax, dtype=np.float64); ax /= np.linalg.norm(ax)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)
    m1 = rot_axis(0.3, [0, 1, 0])
    right = np.array([np.cos(0.3), 0, -np.sin(0.3)])
    m2 = rot_axis(-0.2, right)   # second call: Ry(0) * R_right(x), right from the accumulated yaw
    want = (m2 @ (m1 @ base.T)).T
    assert np.allclose(got, want, atol=2e-5)
    assert np.allclose(c.rot, [-0.2, 0.3, 0.0], atol=1e-6)


def test_camera_pitch_clamps_like_reference():
    c = hostlib.Camera()
    c.rotate(2.0, 0.0, 0.0)   # beyond CAMERA_UPPER_LIMIT 1.570 (Constants.java:9-10)
    assert np.isclose(c.rot[0], 1.570)
    c.rotate(-4.0, 0.0, 0.0)
    assert np.isclose(c.rot[0], -1.570)


def test_camera_strafe_and_pick():
    c = hostlib.Camera()
    c.setPos(1.5, 1.5, 1.5)
    c.setSpeed(0.01)
    c.strafe(1, 0)      # forward = -dir
    assert np.allclose(c.getUniform()[:3], [1.5, 1.5, 1.49])
    c.strafe(0, 2)
    assert np.allclose(c.getUniform()[:3], [1.52, 1.5, 1.49])
    p = c.getRayPickLocation(0.25)   # (-dir * depth + pos - 1) * 8196
    assert list(p) == [int((1.52 - 1) * 8196), int(0.5 * 8196), int(np.float32(np.float32(-0.25) + np.float32(1.49) - 1) * 8196)]


def test_octree_encoders_produce_reference_layout(tmp_path):
    o = hostlib.Octree(64)
    root = o.createInteriorNode(1)
    kids = [o.createSurfaceLeafNode(2, 595), o.createNonSurfaceLeafNode(0), o.createSubdividableLeafNode(1),
            o.createInteriorNode(3)] + [o.createNonSurfaceLeafNode(0) for _ in range(4)]
    o.setChildPointer(root, kids[0])
    mask = 1 | (3 << 2) | (2 << 4) | (0 << 6) | sum(3 << (2 * n) for n in range(4, 8))
    o.setLeafMask(root, mask)
    b = o.getByteBuffer()
    assert o.memOffset == 7 + 3 + 1 + 7 + 7 + 4
    assert list(b[:7]) == [1, 0, 0, 0, 7, mask >> 8, mask & 0xFF]          # value, BE child pointer, BE leaf mask
    assert list(b[7:10]) == [2, 595 & 0xFF, 595 >> 8]                        # surface leaf: LE packed normal
    assert o.getChildPointer(root) == 7 and o.getLeafMask(root) == mask
    rc, st, depth = scene.validate_pool(b)
    assert rc == 0 and st["surface_leaf"] == 1 and st["nonsurface_leaf"] == 5 and st["subdiv_leaf"] == 1


def test_svo_file_roundtrip(tmp_path):
    pool, _ = scene.build_scene(32)
    o = hostlib.Octree(64)
    o.adopt(pool)
    path = str(tmp_path / "debug.svo")
    o.writeBufferToFile(path)
    raw = open(path, "rb").read()
    assert int.from_bytes(raw[:4], "big") == pool.size          # Octree.java:981-985
    assert raw[4:] == pool.tobytes()
    o2 = hostlib.Octree(64)
    o2.readBufferFromFile(path)
    assert o2.memOffset == pool.size and (o2.getByteBuffer() == pool).all()


@pytest.mark.gpu
def test_renderer_mirror_drives_one_frame_like_main():
    """Main.preRun + updateEarly through the C++ Renderer mirror == the oracle."""
    from oracle import oracle
    pool, _ = scene.build_scene(128)
    o = hostlib.Octree(1024)
    o.adopt(pool)
    c = hostlib.Camera()
    c.setPos(1.5, 1.42, 1.5)
    c.rotate(0.0, 0.3, 0.0)
    c.rotate(-0.5, 0.0, 0.0)
    for mode in (2, 0):
        rgba, depth = hostlib.render_frame(o, c, 200, 120, 2, mode)
        ref = oracle.render(pool, 200, 120, c.getUniform(), 2, mode)
        assert (rgba == ref["rgba"]).all()
        assert (depth.view(np.uint32) == ref["depth"].view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("move", [True, False])
def test_renderer_mirror_runs_main_s_loop_with_the_crosshair_pick(move):
    """Main.updateEarly for 24 frames written against the C++ mirror (host/svo_host.cpp::svoh_render_loop): the crosshair depth of
    every frame -- read one turn later, as Main reads it, from the mail of the frame's pick launch while up to four frames are
    in flight -- and the last frame's images == the oracle with that frame's camera and frameNumber."""
    from oracle import oracle
    pool, _ = scene.build_scene(256)
    o = hostlib.Octree(4096)
    o.adopt(pool)
    c = hostlib.Camera()
    c.setPos(1.5, 1.42, 1.5)
    c.setSpeed(1.0)
    c.rotate(-0.5, 0.3, 0.0)
    w, h, n = 320, 184, 24
    picks, cams, fns, rgba, depth = hostlib.render_loop(o, c, w, h, n, 0, move)
    assert (fns == 1).all() if move else (fns == np.arange(2, 2 + n)).all()      # Main.java:225-233, 275
    for i in range(n):
        ref = oracle.render(pool, w, h, cams[i], int(fns[i]), 0, rows=(h // 2, h // 2 + 1), want_hits=False)
        assert ref["depth"].view(np.uint32)[h // 2, w // 2] == picks.view(np.uint32)[i], i
    ref = oracle.render(pool, w, h, cams[-1], int(fns[-1]), 0, want_hits=False)
    assert (rgba == ref["rgba"]).all() and (depth.view(np.uint32) == ref["depth"].view(np.uint32)).all()
