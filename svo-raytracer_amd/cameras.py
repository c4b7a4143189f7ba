"""Benchmark / test cameras: the 15 floats pos, l1, l2, r1, r2."""
import numpy as np

L1, L2, R1, R2 = (-1.6, -0.9, -1.0), (-1.6, 0.9, -1.0), (1.6, -0.9, -1.0), (1.6, 0.9, -1.0)


def rot_cam(pos, pitch, yaw):
    cx, sx, cy, sy = np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    m = ry @ rx
    v = [np.asarray(pos, dtype=np.float64)] + [m @ np.asarray(c) for c in (L1, L2, R1, R2)]
    return np.concatenate(v).astype(np.float32)


CAMERAS = {
    "K0": rot_cam((1.5, 1.5, 2.0), 0.0, 0.0),       # reference default (Main.java:120, Camera.java:13-18)
    "K1": rot_cam((1.5, 1.42, 1.5), -0.5, 0.3),     # inside the cube, pitched toward the terrain
    "K2": rot_cam((1.2, 1.40, 1.8), -0.08, 0.7),    # grazing view
}


def orbit_path(n, start="K1", yaw_step=0.004, pitch_step=0.0, forward=0.0004, side=0.0002):
    """A deterministic moving-camera sequence through the host mirror of the reference's Camera (host/svo_host.hpp <->
    Camera.java:46-50, 76-140): from camera `start`'s position, n frames of rotate(pitch_step, yaw_step, 0) + strafe(forward,
    side) -- what Main.updateEarly does on key / mouse input (Main.java:161-236).  Returns (cams float32 [n][15],
    frame_numbers int32 [n]): every frame moved the camera, so Main resets frameNumber to 0 and pre-increments it: 1
    (Main.java:225-233, 275)."""
    from . import hostlib
    cam = hostlib.Camera()
    p = CAMERAS[start][:3] if isinstance(start, str) else np.asarray(start, dtype=np.float32)[:3]     # a camera's name, or a position
    cam.setPos(float(p[0]), float(p[1]), float(p[2]))
    cam.setSpeed(1.0)
    cam.rotate(-0.5, 0.3, 0.0)     # K1's attitude, through the reference's own rotate()
    out = np.zeros((n, 15), dtype=np.float32)
    for i in range(n):
        cam.rotate(pitch_step, yaw_step, 0.0)
        cam.strafe(forward, side)
        out[i] = cam.getUniform()
    return out, np.ones(n, dtype=np.int32)


def cave_camera(n=8192, seed=1, amp=8, dens=None):
    """A camera INSIDE the largest cave of the "caves" scene (scene.cave_position), looking along the terrain (pitch -0.1, yaw
    0.7): the one view of the matrix from which the scene is concave everywhere -- walls all around, the sky through the cave's
    mouth.  Not one of SURVEY 8(d)'s cameras; `--camera CAVE` of bench.py and tools/matrix.py, with --scene caves."""
    from . import scene
    pos, _ = scene.cave_position(n, seed, amp, scene.CAVES_DENS if dens is None else dens)
    return rot_cam(pos, -0.1, 0.7)
