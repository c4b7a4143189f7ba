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
