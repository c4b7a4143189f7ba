"""ctypes loader for the CPU oracle (oracle/svo_oracle.c).  TEST INFRASTRUCTURE:
imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsvooracle.so")
_lib = None

HIT_DTYPE = np.dtype([("pointer", "<u4"), ("raw_normal", "<u2"), ("value", "u1"), ("level", "u1"),
                      ("iter", "<u4"), ("t", "<f4")])
assert HIT_DTYPE.itemsize == 16


class Params(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("cam", ctypes.c_float * 15),
                ("frame_number", ctypes.c_int32), ("render_mode", ctypes.c_int32), ("buffer_end", ctypes.c_int32),
                ("use_beam", ctypes.c_int32), ("bounces", ctypes.c_int32), ("mirror_mask", ctypes.c_uint32),
                ("spp", ctypes.c_int32), ("progressive", ctypes.c_int32)]


class Stats(ctypes.Structure):
    _fields_ = [("pixels", ctypes.c_uint64), ("rays", ctypes.c_uint64), ("nan_rays", ctypes.c_uint64),
                ("iterations", ctypes.c_uint64), ("alg_bytes", ctypes.c_uint64), ("max_iter", ctypes.c_uint64), ("descends", ctypes.c_uint64),
                ("advances", ctypes.c_uint64), ("pops", ctypes.c_uint64),
                ("push_by_scale", ctypes.c_uint64 * 24), ("pop_by_scale", ctypes.c_uint64 * 24),
                ("cold_pops", ctypes.c_uint64)]

    def as_dict(self):
        return {k: (int(getattr(self, k)) if not hasattr(getattr(self, k), '__len__') else list(getattr(self, k)))
                for k, _ in self._fields_}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libsvooracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.svo_oracle_render.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(Params), ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.POINTER(Stats)]
        L.svo_oracle_render.restype = ctypes.c_int
        L.svo_oracle_render_beam.argtypes = L.svo_oracle_render.argtypes + [ctypes.c_void_p]
        L.svo_oracle_render_beam.restype = ctypes.c_int
        L.svo_oracle_render_mt.argtypes = L.svo_oracle_render.argtypes + [ctypes.c_int]
        L.svo_oracle_render_mt.restype = ctypes.c_int
        L.svo_oracle_beam.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(Params), ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_uint64)]
        L.svo_oracle_beam.restype = ctypes.c_int
        for name in ("sin", "cos", "acos", "exp2"):
            f = getattr(L, "svo_oracle_" + name)
            f.argtypes = [ctypes.c_float]
            f.restype = ctypes.c_float
        L.svo_oracle_rand.argtypes = [ctypes.c_float, ctypes.c_float]
        L.svo_oracle_rand.restype = ctypes.c_float
        _lib = L
    return _lib


def _params(pool, width, height, cam, frame_number, render_mode, bounces, mirror_mask, spp, progressive=0):
    prm = Params()
    prm.width, prm.height = int(width), int(height)
    cam = np.asarray(cam, dtype=np.float32).reshape(15)
    for i in range(15):
        prm.cam[i] = float(cam[i])
    prm.frame_number, prm.render_mode = int(frame_number), int(render_mode)
    prm.buffer_end, prm.use_beam = int(pool.size), 0
    prm.bounces, prm.mirror_mask, prm.spp = int(bounces), int(mirror_mask), int(spp)
    prm.progressive = int(progressive)
    return prm


BEAM_BLOCK = 4   # Main.java:41 beamSquareSize


def beam(pool, width, height, cam, want_visits=False):
    """The coarse pass of use_beam = 1: conservative start distance per 4x4 pixel block, float32 [ceil(H/4)][ceil(W/4)]."""
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    prm = _params(pool, width, height, cam, 2, 2, 2, 0, 1)
    out = np.zeros(((height + BEAM_BLOCK - 1) // BEAM_BLOCK, (width + BEAM_BLOCK - 1) // BEAM_BLOCK), dtype=np.float32)
    nv = ctypes.c_uint64()
    rc = lib().svo_oracle_beam(pool.ctypes.data, pool.size, ctypes.byref(prm), out.ctypes.data, ctypes.byref(nv))
    if rc != 0:
        raise RuntimeError(f"svo_oracle_beam rc={rc}")
    return (out, int(nv.value)) if want_visits else out


def render(pool, width, height, cam, frame_number=2, render_mode=2, bounces=2, mirror_mask=0, spp=1,
           rows=None, xstep=1, ystep=1, want_hits=True, use_beam=False, last_rgba=None, threads=1):
    """Run the CPU restatement. cam: 15 floats (pos,l1,l2,r1,r2). Returns dict with
    rgba (H,W,4 u8), depth (H,W f32), hits (H,W HIT_DTYPE), stats.  use_beam: primary rays start at the coarse
    pass's distance of their block (see svo_oracle_beam in svo_oracle.c); the result also carries "beam"."""
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    prm = _params(pool, width, height, cam, frame_number, render_mode, bounces, mirror_mask, spp,
                  progressive=last_rgba is not None)
    tb = beam(pool, width, height, cam) if use_beam else None
    y0, y1 = (0, height) if rows is None else rows
    # last_rgba: the image the previous frame left (cross-frame accumulation, svotrace.comp:712-719, dormant)
    rgba = np.zeros((height, width, 4), dtype=np.uint8) if last_rgba is None else np.array(last_rgba, dtype=np.uint8).reshape(height, width, 4).copy()
    depth = np.zeros((height, width), dtype=np.float32)
    hits = np.zeros((height, width), dtype=HIT_DTYPE) if want_hits else None
    st = Stats()
    if threads > 1 and tb is None and last_rgba is None:   # all host cores (the CPU figure beside the single-thread one)
        rc = lib().svo_oracle_render_mt(pool.ctypes.data, pool.size, ctypes.byref(prm), int(y0), int(y1), int(xstep),
                                        int(ystep), rgba.ctypes.data, depth.ctypes.data,
                                        hits.ctypes.data if want_hits else None, ctypes.byref(st), int(threads))
    else:
        rc = lib().svo_oracle_render_beam(pool.ctypes.data, pool.size, ctypes.byref(prm), int(y0), int(y1), int(xstep),
                                          int(ystep), rgba.ctypes.data, depth.ctypes.data,
                                          hits.ctypes.data if want_hits else None, ctypes.byref(st),
                                          tb.ctypes.data if tb is not None else None)
    if rc != 0:
        raise RuntimeError(f"svo_oracle_render rc={rc}")
    return {"rgba": rgba, "depth": depth, "hits": hits, "stats": st.as_dict(), "beam": tb}
