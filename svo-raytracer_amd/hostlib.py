"""ctypes binding of libsvohost.so: the C++ host mirror of the reference's Java classes
(host/svo_host.hpp: Renderer / Camera / Octree, same names and meaning as the reference)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "host", "libsvohost.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run __graft_entry__.build()")
        from . import one_hip_runtime
        one_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.svoh_camera_new.restype = vp
        for n in ("svoh_camera_free", "svoh_octree_free"):
            getattr(L, n).argtypes = [vp]
        L.svoh_camera_set_pos.argtypes = [vp, cf, cf, cf]
        L.svoh_camera_set_speed.argtypes = [vp, cf]
        L.svoh_camera_rotate.argtypes = [vp, cf, cf, cf]
        L.svoh_camera_strafe.argtypes = [vp, cf, cf]
        for n in ("svoh_camera_get_uniform", "svoh_camera_get_dir", "svoh_camera_get_rot"):
            getattr(L, n).argtypes = [vp, vp]
        L.svoh_camera_pick.argtypes = [vp, cf, vp]
        L.svoh_octree_new.argtypes = [ci]
        L.svoh_octree_new.restype = vp
        L.svoh_octree_adopt.argtypes = [vp, vp, ctypes.c_uint64]
        L.svoh_octree_mem_offset.argtypes = [vp]
        L.svoh_octree_buffer.argtypes = [vp]
        L.svoh_octree_buffer.restype = vp
        L.svoh_octree_write.argtypes = [vp, ctypes.c_char_p]
        L.svoh_octree_read.argtypes = [vp, ctypes.c_char_p]
        for n in ("svoh_octree_create_interior", "svoh_octree_create_nonsurface_leaf",
                  "svoh_octree_create_subdividable_leaf", "svoh_octree_get_child_pointer", "svoh_octree_get_leaf_mask"):
            getattr(L, n).argtypes = [vp, ci]
        L.svoh_octree_create_surface_leaf.argtypes = [vp, ci, ci]
        L.svoh_octree_set_child_pointer.argtypes = [vp, ci, ci]
        L.svoh_octree_set_leaf_mask.argtypes = [vp, ci, ci]
        L.svoh_render_frame.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp]
        _lib = L
    return _lib


class Camera:
    """src/engine/Camera.java"""

    def __init__(self):
        self._L = lib()
        self._h = self._L.svoh_camera_new()

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.svoh_camera_free(self._h)
            self._h = None

    def setPos(self, x, y, z):
        self._L.svoh_camera_set_pos(self._h, x, y, z)

    def setSpeed(self, s):
        self._L.svoh_camera_set_speed(self._h, s)

    def rotate(self, x, y, z):
        self._L.svoh_camera_rotate(self._h, x, y, z)

    def strafe(self, forward, side):
        self._L.svoh_camera_strafe(self._h, forward, side)

    def _get(self, fn, n, dtype=np.float32):
        out = np.zeros(n, dtype=dtype)
        fn(self._h, out.ctypes.data)
        return out

    def getUniform(self):
        return self._get(self._L.svoh_camera_get_uniform, 15)

    @property
    def dir(self):
        return self._get(self._L.svoh_camera_get_dir, 3)

    @property
    def rot(self):
        return self._get(self._L.svoh_camera_get_rot, 3)

    def getRayPickLocation(self, depth):
        out = np.zeros(3, dtype=np.int32)
        self._L.svoh_camera_pick(self._h, depth, out.ctypes.data)
        return out


class Octree:
    """src/engine/Octree.java (pool, encoders, .svo IO)"""

    def __init__(self, memSizeKB):
        self._L = lib()
        self._h = self._L.svoh_octree_new(int(memSizeKB))
        self.bufferSize = int(memSizeKB) * 1024

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.svoh_octree_free(self._h)
            self._h = None

    @property
    def memOffset(self):
        return self._L.svoh_octree_mem_offset(self._h)

    def adopt(self, pool):
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        self._L.svoh_octree_adopt(self._h, pool.ctypes.data, pool.size)

    def getByteBuffer(self):
        n = self.memOffset
        p = self._L.svoh_octree_buffer(self._h)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)).copy()

    def writeBufferToFile(self, path):
        self._L.svoh_octree_write(self._h, path.encode())

    def readBufferFromFile(self, path):
        self._L.svoh_octree_read(self._h, path.encode())

    def createInteriorNode(self, v):
        return self._L.svoh_octree_create_interior(self._h, v)

    def createSurfaceLeafNode(self, v, normal):
        return self._L.svoh_octree_create_surface_leaf(self._h, v, normal)

    def createNonSurfaceLeafNode(self, v):
        return self._L.svoh_octree_create_nonsurface_leaf(self._h, v)

    def createSubdividableLeafNode(self, v):
        return self._L.svoh_octree_create_subdividable_leaf(self._h, v)

    def setChildPointer(self, parent, child):
        self._L.svoh_octree_set_child_pointer(self._h, parent, child)

    def getChildPointer(self, parent):
        return self._L.svoh_octree_get_child_pointer(self._h, parent)

    def setLeafMask(self, parent, mask):
        self._L.svoh_octree_set_leaf_mask(self._h, parent, mask)

    def getLeafMask(self, parent):
        return self._L.svoh_octree_get_leaf_mask(self._h, parent)


def render_frame(octree, camera, width, height, frame_number=2, render_mode=2):
    """Main.preRun + one Main.updateEarly through the C++ Renderer mirror (needs a GPU)."""
    rgba = np.zeros((height, width, 4), dtype=np.uint8)
    depth = np.zeros((height, width), dtype=np.float32)
    rc = lib().svoh_render_frame(octree._h, camera._h, width, height, frame_number, render_mode, rgba.ctypes.data,
                                 depth.ctypes.data)
    if rc != 0:
        raise RuntimeError("Renderer reported an error (see stdout)")
    return rgba, depth


def render_loop(octree, camera, width, height, nframes, render_mode=0, move=True):
    """Main.updateEarly for `nframes` frames through the C++ Renderer mirror (needs a GPU): returns (picks float32 [n] -- the
    crosshair depth of every frame, read the way Main reads it, one turn later --, cams float32 [n][15], frame numbers int32 [n],
    the last frame's rgba8 and depth images)."""
    picks = np.zeros(nframes, dtype=np.float32)
    cams = np.zeros((nframes, 15), dtype=np.float32)
    fns = np.zeros(nframes, dtype=np.int32)
    rgba = np.zeros((height, width, 4), dtype=np.uint8)
    depth = np.zeros((height, width), dtype=np.float32)
    L = lib()
    L.svoh_render_loop.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 5
    L.svoh_render_loop.restype = ctypes.c_int
    rc = L.svoh_render_loop(octree._h, camera._h, width, height, nframes, render_mode, 1 if move else 0, picks.ctypes.data,
                            cams.ctypes.data, fns.ctypes.data, rgba.ctypes.data, depth.ctypes.data)
    if rc != 0:
        raise RuntimeError("Renderer reported an error (see stdout)")
    return picks, cams, fns, rgba, depth
