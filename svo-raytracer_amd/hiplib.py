"""ctypes binding of libsvohip.so (include/svo_hip.h).  Thin: one Python method per
C-ABI entry point.  Fails loudly when the HIP library is missing -- there is no CPU
fallback on the product path."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVO_HIP_LIB") or os.path.join(_HERE, "csrc", "libsvohip.so")  # SVO_HIP_LIB: A/B builds

HIT_DTYPE = np.dtype([("pointer", "<u4"), ("raw_normal", "<u2"), ("value", "u1"), ("level", "u1"),
                      ("iter", "<u4"), ("t", "<f4")])

EXPORTS = [
    "svo_build_info", "svo_create", "svo_destroy", "svo_last_error", "svo_pool_upload", "svo_pool_update", "svo_pool_download",
    "svo_pool_reserve", "svo_pool_upload_device", "svo_pool_device_ptr", "svo_build_from_heightmap", "svo_build_from_voxels", "svo_bind_outputs", "svo_set_camera", "svo_set_params", "svo_resize", "svo_set_rows", "svo_set_stripes",
    "svo_set_pipeline", "svo_set_tuning", "svo_set_hit_records", "svo_set_progressive", "svo_set_batch", "svo_dispatch", "svo_dispatch_async", "svo_sync", "svo_set_pick", "svo_set_overlap", "svo_pick_info", "svo_count_frame",
    "svo_get_stats", "svo_set_stream", "svo_time_frames", "svo_read_color", "svo_read_depth", "svo_read_hits", "svo_read_pixel", "svo_read_beam",
    "svo_output_device_ptrs", "svo_set_derived", "svo_derived_info", "svo_derived_refresh_info",
    "svo_ring_create", "svo_ring_destroy", "svo_ring_submit", "svo_ring_wait", "svo_ring_query", "svo_ring_read_color",
    "svo_ring_read_depth", "svo_ring_read_hits", "svo_ring_read_pixel", "svo_ring_bind_slot", "svo_ring_device_ptrs",
    "svo_set_reserved_cus", "svo_pool_commit", "svo_ring_forward_slot", "svo_dev_alloc", "svo_dev_free", "svo_dev_read",
    "svo_ipc_export", "svo_ipc_open", "svo_ipc_close", "svo_set_sequence", "svo_ring_submit_cams",
    "svo_build_from_heightmap16", "svo_launch_info",
    "svo_group_create", "svo_group_destroy", "svo_group_last_error", "svo_group_size", "svo_group_member", "svo_group_pool_upload",
    "svo_group_pool_update", "svo_group_pool_download", "svo_group_build_from_heightmap", "svo_group_set_camera",
    "svo_group_set_params", "svo_group_set_pipeline", "svo_group_set_tuning", "svo_group_set_progressive", "svo_group_set_sequence",
    "svo_group_resize", "svo_group_ring_create", "svo_group_ring_destroy", "svo_group_ring_submit", "svo_group_ring_submit_cams",
    "svo_group_ring_wait", "svo_group_ring_query", "svo_group_ring_read_color", "svo_group_ring_read_depth",
    "svo_group_ring_read_hits", "svo_group_ring_read_pixel",
]


class Stats(ctypes.Structure):
    _fields_ = [("pixels", ctypes.c_uint64), ("rays", ctypes.c_uint64), ("nan_rays", ctypes.c_uint64),
                ("iterations", ctypes.c_uint64), ("alg_bytes", ctypes.c_uint64), ("max_iter", ctypes.c_uint64),
                ("last_dispatch_ms", ctypes.c_float), ("device", ctypes.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# the same library built with hipcc's translation of the traversal loop instead of the assembly one
# (csrc/Makefile target `cxxloop`); only the cross-check test loads it
CXXLOOP_LIB_PATH = os.path.join(_HERE, "csrc", "libsvohip_cxxloop.so")
# the product library's sources with the comparators and A/B switches built in (csrc/Makefile target `variants`): pipeline 2
# (staged wavefront tracing), round 5's spare-ray kernel (SVO_SPARE=1) and the SVO_* environment switches.  Loaded by the tests
# that compare against them (tests/helpers.py::DualContext) and by the tools/ A/B scripts; a host loads libsvohip.so.
VARIANTS_LIB_PATH = os.path.join(_HERE, "csrc", "libsvohip_variants.so")

_libs = {}


def lib(path=None):
    path = path or LIB_PATH
    _lib = _libs.get(path)
    if _lib is None:
        if not os.path.exists(path):
            raise RuntimeError(f"HIP library missing: {path} (run __graft_entry__.build()); "
                               "the SVO hot path has no CPU fallback")
        from . import one_hip_runtime
        one_hip_runtime()
        L = ctypes.CDLL(path)
        vp, u64, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int
        fp = ctypes.POINTER(ctypes.c_float)
        L.svo_create.argtypes = [ci, ctypes.POINTER(vp)]
        L.svo_destroy.argtypes = [vp]
        L.svo_last_error.argtypes = [vp]
        L.svo_last_error.restype = ctypes.c_char_p
        L.svo_pool_upload.argtypes = [vp, vp, u64]
        L.svo_pool_update.argtypes = [vp, vp, u64, u64]
        L.svo_pool_download.argtypes = [vp, vp, u64]
        L.svo_pool_reserve.argtypes = [vp, u64]
        L.svo_pool_upload_device.argtypes = [vp, vp, u64]
        L.svo_build_from_heightmap.argtypes = [vp, vp, vp, ci, ctypes.POINTER(u64)]
        L.svo_build_from_voxels.argtypes = [vp, vp, ci, ctypes.POINTER(u64)]
        L.svo_build_from_heightmap16.argtypes = [vp, vp, vp, ci, ctypes.POINTER(u64)]
        L.svo_bind_outputs.argtypes = [vp, vp, vp, vp]
        L.svo_pool_device_ptr.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u64)]
        L.svo_set_camera.argtypes = [vp, fp, fp, fp, fp, fp]
        L.svo_set_params.argtypes = [vp, ci, ci, ci, ci, ci, ctypes.c_uint32, ci]
        L.svo_resize.argtypes = [vp, ci, ci]
        L.svo_set_rows.argtypes = [vp, ci, ci]
        L.svo_set_stripes.argtypes = [vp, ci, ci, ci, ci]
        L.svo_set_pipeline.argtypes = [vp, ci]
        L.svo_set_hit_records.argtypes = [vp, ci]
        L.svo_set_progressive.argtypes = [vp, ci]
        L.svo_set_batch.argtypes = [vp, ci, u64]
        L.svo_set_tuning.argtypes = [vp, ci, ci]
        ip = ctypes.POINTER(ci)
        L.svo_ring_create.argtypes = [vp, ci, ci, ci]
        L.svo_ring_destroy.argtypes = [vp]
        L.svo_set_reserved_cus.argtypes = [vp, ci]
        L.svo_pool_commit.argtypes = [vp]
        L.svo_ring_forward_slot.argtypes = [vp, ci, vp, vp, u64, vp]
        L.svo_dev_alloc.argtypes = [vp, u64, ctypes.POINTER(vp)]
        L.svo_dev_free.argtypes = [vp, vp]
        L.svo_dev_read.argtypes = [vp, vp, vp, u64]
        L.svo_ipc_export.argtypes = [vp, vp, vp]
        L.svo_ipc_open.argtypes = [vp, vp, ctypes.POINTER(vp)]
        L.svo_ipc_close.argtypes = [vp, vp]
        L.svo_ring_submit.argtypes = [vp, ci, ci, ip]
        L.svo_ring_submit_cams.argtypes = [vp, ci, vp, vp, ip]
        L.svo_set_sequence.argtypes = [vp, ci, ci]
        L.svo_ring_wait.argtypes = [vp, ci]
        L.svo_ring_query.argtypes = [vp, ci, ip, ip, ip, fp]
        L.svo_ring_read_color.argtypes = [vp, ci, ci, vp]
        L.svo_ring_read_depth.argtypes = [vp, ci, ci, vp]
        L.svo_ring_read_hits.argtypes = [vp, ci, ci, vp]
        L.svo_ring_read_pixel.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp]
        L.svo_ring_bind_slot.argtypes = [vp, ci, vp, vp, vp, u64]
        L.svo_ring_device_ptrs.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                           ctypes.POINTER(u64), ctypes.POINTER(vp)]
        L.svo_set_derived.argtypes = [vp, ci]
        L.svo_derived_info.argtypes = [vp, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(ci), fp]
        L.svo_derived_refresh_info.argtypes = [vp, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64), fp]
        L.svo_build_info.argtypes = []
        L.svo_dispatch.argtypes = [vp]
        L.svo_dispatch_async.argtypes = [vp]
        L.svo_sync.argtypes = [vp]
        L.svo_set_pick.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        L.svo_set_overlap.argtypes = [vp, ctypes.c_int]
        L.svo_pick_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64),
                                    ctypes.POINTER(ctypes.c_uint64)]
        L.svo_count_frame.argtypes = [vp, ctypes.POINTER(Stats)]
        L.svo_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
        L.svo_set_stream.argtypes = [vp, vp]
        L.svo_time_frames.argtypes = [vp, ci, ci, fp]
        L.svo_read_color.argtypes = [vp, vp]
        L.svo_read_depth.argtypes = [vp, vp]
        L.svo_read_hits.argtypes = [vp, vp]
        L.svo_read_beam.argtypes = [vp, vp]
        L.svo_read_pixel.argtypes = [vp, ci, ci, vp, vp, vp]
        L.svo_output_device_ptrs.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
        # N GPUs behind the boundary (svo_group_*)
        L.svo_group_create.argtypes = [ip, ci, ctypes.POINTER(vp)]
        L.svo_group_destroy.argtypes = [vp]
        L.svo_group_last_error.argtypes = [vp]
        L.svo_group_last_error.restype = ctypes.c_char_p
        L.svo_group_size.argtypes = [vp]
        L.svo_group_member.argtypes = [vp, ci]
        L.svo_group_member.restype = vp
        L.svo_group_pool_upload.argtypes = [vp, vp, u64]
        L.svo_group_pool_update.argtypes = [vp, vp, u64, u64]
        L.svo_group_pool_download.argtypes = [vp, vp, u64]
        L.svo_group_build_from_heightmap.argtypes = [vp, vp, vp, ci, ctypes.POINTER(u64)]
        L.svo_group_set_camera.argtypes = [vp, fp, fp, fp, fp, fp]
        L.svo_group_set_params.argtypes = [vp, ci, ci, ci, ci, ci, ctypes.c_uint32, ci]
        L.svo_group_set_pipeline.argtypes = [vp, ci]
        L.svo_group_set_tuning.argtypes = [vp, ci, ci]
        L.svo_group_set_progressive.argtypes = [vp, ci]
        L.svo_group_set_sequence.argtypes = [vp, ci, ci]
        L.svo_group_resize.argtypes = [vp, ci, ci]
        L.svo_group_ring_create.argtypes = [vp, ci, ci, ci, ci]
        L.svo_group_ring_destroy.argtypes = [vp]
        L.svo_group_ring_submit.argtypes = [vp, ci, ci, ip]
        L.svo_group_ring_submit_cams.argtypes = [vp, ci, vp, vp, ip]
        L.svo_group_ring_wait.argtypes = [vp, ci]
        L.svo_group_ring_query.argtypes = [vp, ci, ip, ip, ip, fp]
        L.svo_group_ring_read_color.argtypes = [vp, ci, ci, vp]
        L.svo_group_ring_read_depth.argtypes = [vp, ci, ci, vp]
        L.svo_group_ring_read_hits.argtypes = [vp, ci, ci, vp]
        L.svo_group_ring_read_pixel.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp]
        for n in EXPORTS:
            if n not in ("svo_last_error", "svo_group_last_error", "svo_group_member"):
                getattr(L, n).restype = ci
        _lib = _libs[path] = L
    return _lib


class SvoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"svo error {code}: {msg}")
        self.code = code


class HipGroup:
    """N GPUs behind the C ABI (include/svo_hip.h, svo_group_*): one process, one thread; member r renders every n-th tile
    row, member 0 owns the assembled frames.  `devices` may repeat a device (tests on one GPU)."""

    def __init__(self, devices, lib_path=None):
        self._L = lib(lib_path)
        self._h = ctypes.c_void_p()
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._L.svo_group_create(devs, len(devices), ctypes.byref(self._h))
        if rc != 0:
            raise SvoError(rc, "svo_group_create failed (no GPU / bad device index)")
        self.n = len(devices)
        self.width = self.height = 0

    def _chk(self, rc):
        if rc != 0:
            raise SvoError(rc, self._L.svo_group_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._L.svo_group_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def member(self, i):
        """a HipContext view of member i (stats, descriptor table info); not to be closed"""
        c = HipContext.__new__(HipContext)
        c._L = self._L
        c._h = ctypes.c_void_p(self._L.svo_group_member(self._h, int(i)))
        c.width, c.height = self.width, self.height
        c.close = lambda: None
        return c

    def pool_upload(self, pool):
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        self._chk(self._L.svo_group_pool_upload(self._h, pool.ctypes.data, pool.size))

    def pool_update(self, pool, start, end):
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        self._chk(self._L.svo_group_pool_update(self._h, pool.ctypes.data, int(start), int(end)))

    def pool_download(self, nbytes):
        out = np.zeros(int(nbytes), dtype=np.uint8)
        self._chk(self._L.svo_group_pool_download(self._h, out.ctypes.data, out.size))
        return out

    def build_from_heightmap(self, height, material):
        height = np.ascontiguousarray(height, dtype=np.uint16)
        material = np.ascontiguousarray(material, dtype=np.uint8)
        nb = ctypes.c_uint64()
        self._chk(self._L.svo_group_build_from_heightmap(self._h, height.ctypes.data, material.ctypes.data, height.shape[0], ctypes.byref(nb)))
        return int(nb.value)

    def set_camera(self, cam):
        cam = np.ascontiguousarray(np.asarray(cam, dtype=np.float32).reshape(5, 3))
        fp = ctypes.POINTER(ctypes.c_float)
        self._chk(self._L.svo_group_set_camera(self._h, *[cam[i].ctypes.data_as(fp) for i in range(5)]))

    def set_params(self, frame_number=2, render_mode=2, buffer_end=0, use_beam=0, bounces=2, mirror_mask=0, spp=1):
        self._chk(self._L.svo_group_set_params(self._h, int(frame_number), int(render_mode), int(buffer_end), int(use_beam),
                                               int(bounces), int(mirror_mask), int(spp)))

    def set_pipeline(self, p):
        self._chk(self._L.svo_group_set_pipeline(self._h, int(p)))

    def set_tuning(self, waves_per_cu=0, round_threshold_sixteenths=0):
        self._chk(self._L.svo_group_set_tuning(self._h, int(waves_per_cu), int(round_threshold_sixteenths)))

    def set_progressive(self, on):
        self._chk(self._L.svo_group_set_progressive(self._h, 1 if on else 0))

    def set_sequence(self, nframes, fresh=True):
        self._chk(self._L.svo_group_set_sequence(self._h, int(nframes), 1 if fresh else 0))

    def resize(self, width, height):
        self._chk(self._L.svo_group_resize(self._h, int(width), int(height)))
        self.width, self.height = int(width), int(height)

    def ring_create(self, slots, frames_per_slot=1, want_hits=False, exchange=0):
        self._chk(self._L.svo_group_ring_create(self._h, int(slots), int(frames_per_slot), 1 if want_hits else 0, int(exchange)))

    def ring_destroy(self):
        self._chk(self._L.svo_group_ring_destroy(self._h))

    def ring_submit(self, frame_number, nframes=1):
        slot = ctypes.c_int()
        self._chk(self._L.svo_group_ring_submit(self._h, int(frame_number), int(nframes), ctypes.byref(slot)))
        return int(slot.value)

    def ring_submit_cams(self, cams, frame_numbers):
        cams = np.ascontiguousarray(np.asarray(cams, dtype=np.float32).reshape(-1, 15))
        fn = np.ascontiguousarray(np.asarray(frame_numbers, dtype=np.int32).reshape(-1))
        slot = ctypes.c_int()
        self._chk(self._L.svo_group_ring_submit_cams(self._h, int(fn.size), cams.ctypes.data, fn.ctypes.data, ctypes.byref(slot)))
        return int(slot.value)

    def ring_wait(self, slot):
        self._chk(self._L.svo_group_ring_wait(self._h, int(slot)))

    def ring_query(self, slot):
        done, first, n, ms = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_float()
        self._chk(self._L.svo_group_ring_query(self._h, int(slot), ctypes.byref(done), ctypes.byref(first), ctypes.byref(n), ctypes.byref(ms)))
        return {"done": bool(done.value), "first_frame": int(first.value), "nframes": int(n.value), "gpu_ms": float(ms.value)}

    def ring_read(self, slot, k=0, want_hits=False):
        out = {"rgba": np.zeros((self.height, self.width, 4), dtype=np.uint8),
               "depth": np.zeros((self.height, self.width), dtype=np.float32)}
        self._chk(self._L.svo_group_ring_read_color(self._h, int(slot), int(k), out["rgba"].ctypes.data))
        self._chk(self._L.svo_group_ring_read_depth(self._h, int(slot), int(k), out["depth"].ctypes.data))
        if want_hits:
            out["hits"] = np.zeros((self.height, self.width), dtype=HIT_DTYPE)
            self._chk(self._L.svo_group_ring_read_hits(self._h, int(slot), int(k), out["hits"].ctypes.data))
        return out

    def ring_read_pixel(self, slot, k, x, y, want_hit=True):
        rgba = np.zeros(4, dtype=np.uint8)
        depth = np.zeros(1, dtype=np.float32)
        hit = np.zeros(1, dtype=HIT_DTYPE)
        self._chk(self._L.svo_group_ring_read_pixel(self._h, int(slot), int(k), int(x), int(y), rgba.ctypes.data, depth.ctypes.data,
                                                    hit.ctypes.data if want_hit else None))
        return rgba, float(depth[0]), hit[0]


class HipContext:
    """One context per GPU (include/svo_hip.h)."""

    def __init__(self, device=0, lib_path=None):
        self._L = lib(lib_path)
        self._h = ctypes.c_void_p()
        rc = self._L.svo_create(int(device), ctypes.byref(self._h))
        if rc != 0:
            raise SvoError(rc, "svo_create failed (no GPU / bad device index)")
        self.width = self.height = 0

    def _chk(self, rc):
        if rc != 0:
            raise SvoError(rc, self._L.svo_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._L.svo_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # pool
    def pool_upload(self, pool):
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        self._chk(self._L.svo_pool_upload(self._h, pool.ctypes.data, pool.size))

    def pool_update(self, pool, start, end):
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        self._chk(self._L.svo_pool_update(self._h, pool.ctypes.data, int(start), int(end)))

    def pool_download(self, nbytes):
        out = np.zeros(int(nbytes), dtype=np.uint8)
        self._chk(self._L.svo_pool_download(self._h, out.ctypes.data, out.size))
        return out

    def pool_upload_device(self, dptr, nbytes):
        self._chk(self._L.svo_pool_upload_device(self._h, ctypes.c_void_p(int(dptr)), int(nbytes)))

    def build_from_heightmap(self, height, material):
        """GPU world generation (svo_build_from_heightmap); returns the size of the new pool."""
        height = np.ascontiguousarray(height, dtype=np.uint16)
        material = np.ascontiguousarray(material, dtype=np.uint8)
        n = height.shape[0]
        assert height.shape == (n, n) and material.shape == (n, n)
        nb = ctypes.c_uint64()
        self._chk(self._L.svo_build_from_heightmap(self._h, height.ctypes.data, material.ctypes.data, n, ctypes.byref(nb)))
        return int(nb.value)

    def build_from_heightmap16(self, raw16, material):
        """the same from raw 16-bit height samples as the reference feeds them to its shader (svo_build_from_heightmap16)"""
        raw16 = np.ascontiguousarray(raw16, dtype=np.uint16)
        material = np.ascontiguousarray(material, dtype=np.uint8)
        n = raw16.shape[0]
        assert raw16.shape == (n, n) and material.shape == (n, n)
        nb = ctypes.c_uint64()
        self._chk(self._L.svo_build_from_heightmap16(self._h, raw16.ctypes.data, material.ctypes.data, n, ctypes.byref(nb)))
        return int(nb.value)

    def build_from_voxels(self, grid):
        """GPU builder over a dense chunk grid[z, y, x] (svo_build_from_voxels); returns the size of the new pool."""
        grid = np.ascontiguousarray(grid, dtype=np.uint8)
        n = grid.shape[0]
        assert grid.shape == (n, n, n)
        nb = ctypes.c_uint64()
        self._chk(self._L.svo_build_from_voxels(self._h, grid.ctypes.data, n, ctypes.byref(nb)))
        return int(nb.value)

    def bind_outputs(self, color_ptr, depth_ptr, hits_ptr=None):
        self._chk(self._L.svo_bind_outputs(self._h, ctypes.c_void_p(color_ptr or 0), ctypes.c_void_p(depth_ptr or 0),
                                           ctypes.c_void_p(hits_ptr or 0)))

    def pool_reserve(self, nbytes):
        self._chk(self._L.svo_pool_reserve(self._h, int(nbytes)))

    def pool_commit(self):
        self._chk(self._L.svo_pool_commit(self._h))

    def pool_device_ptr(self):
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        self._chk(self._L.svo_pool_device_ptr(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    # frame state
    def set_camera(self, cam):
        cam = np.ascontiguousarray(np.asarray(cam, dtype=np.float32).reshape(5, 3))
        fp = ctypes.POINTER(ctypes.c_float)
        ptrs = [cam[i].ctypes.data_as(fp) for i in range(5)]
        self._chk(self._L.svo_set_camera(self._h, *ptrs))

    def set_params(self, frame_number=2, render_mode=2, buffer_end=0, use_beam=0, bounces=2, mirror_mask=0, spp=1):
        self._chk(self._L.svo_set_params(self._h, int(frame_number), int(render_mode), int(buffer_end), int(use_beam),
                                         int(bounces), int(mirror_mask), int(spp)))

    def resize(self, width, height):
        self._chk(self._L.svo_resize(self._h, int(width), int(height)))
        self.width, self.height = int(width), int(height)

    def set_rows(self, y0, y1):
        self._chk(self._L.svo_set_rows(self._h, int(y0), int(y1)))

    def set_stripes(self, first_tile_row, tile_row_step, n_tile_rows, out_row0):
        self._chk(self._L.svo_set_stripes(self._h, int(first_tile_row), int(tile_row_step), int(n_tile_rows), int(out_row0)))

    def set_pipeline(self, p):
        self._chk(self._L.svo_set_pipeline(self._h, int(p)))

    def set_tuning(self, waves_per_cu=0, round_threshold_sixteenths=0):
        self._chk(self._L.svo_set_tuning(self._h, int(waves_per_cu), int(round_threshold_sixteenths)))

    def launch_info(self):
        """shape of the last pipeline-1 launch: what the automatic launch shape (svo_set_tuning's 0) resolved to"""
        w, p, t = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        self._chk(self._L.svo_launch_info(self._h, ctypes.byref(w), ctypes.byref(p), ctypes.byref(t)))
        return {"waves": w.value, "waves_per_cu": p.value, "round_threshold_sixteenths": t.value}

    def set_derived(self, mode):
        """0 = walk the pool's records as the shader does, 1 (default) = walk the interior-descriptor table when the
        pool is derivable."""
        self._chk(self._L.svo_set_derived(self._h, int(mode)))

    def derived_info(self):
        n, b, w, ms = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int(), ctypes.c_float()
        self._chk(self._L.svo_derived_info(self._h, ctypes.byref(n), ctypes.byref(b), ctypes.byref(w), ctypes.byref(ms)))
        return {"descriptors": int(n.value), "bytes": int(b.value), "walkable": bool(w.value), "build_ms": float(ms.value)}

    def derived_refresh_info(self):
        """what svo_pool_update did to the table: updates followed without a rebuild; states recomputed, descriptors appended
        and GPU ms of the last one"""
        r, n, a, ms = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_float()
        self._chk(self._L.svo_derived_refresh_info(self._h, ctypes.byref(r), ctypes.byref(n), ctypes.byref(a), ctypes.byref(ms)))
        return {"refreshes": int(r.value), "states": int(n.value), "added": int(a.value), "gpu_ms": float(ms.value)}

    def set_batch(self, nframes, frame_stride=0):
        self._chk(self._L.svo_set_batch(self._h, int(nframes), int(frame_stride)))

    def set_progressive(self, on):
        self._chk(self._L.svo_set_progressive(self._h, 1 if on else 0))

    def set_sequence(self, nframes, fresh=True):
        """progressive: frames of the cross-frame accumulation per dispatch (svo_set_sequence)"""
        self._chk(self._L.svo_set_sequence(self._h, int(nframes), 1 if fresh else 0))

    def set_hit_records(self, on):
        self._chk(self._L.svo_set_hit_records(self._h, 1 if on else 0))

    def set_stream(self, stream_ptr):
        self._chk(self._L.svo_set_stream(self._h, ctypes.c_void_p(stream_ptr or 0)))

    # frames in flight behind the boundary (svo_ring_*)
    def ring_create(self, slots, frames_per_slot=1, want_hits=False):
        self._chk(self._L.svo_ring_create(self._h, int(slots), int(frames_per_slot), 1 if want_hits else 0))

    def set_reserved_cus(self, per_xcd):
        self._chk(self._L.svo_set_reserved_cus(self._h, int(per_xcd)))

    def ring_destroy(self):
        self._chk(self._L.svo_ring_destroy(self._h))

    def ring_submit(self, frame_number, nframes=1):
        slot = ctypes.c_int()
        self._chk(self._L.svo_ring_submit(self._h, int(frame_number), int(nframes), ctypes.byref(slot)))
        return int(slot.value)

    def ring_submit_cams(self, cams, frame_numbers):
        """one submission whose frames carry their own camera (15 floats each) and frameNumber (svo_ring_submit_cams)"""
        cams = np.ascontiguousarray(np.asarray(cams, dtype=np.float32).reshape(-1, 15))
        fn = np.ascontiguousarray(np.asarray(frame_numbers, dtype=np.int32).reshape(-1))
        assert cams.shape[0] == fn.size
        slot = ctypes.c_int()
        self._chk(self._L.svo_ring_submit_cams(self._h, int(fn.size), cams.ctypes.data, fn.ctypes.data, ctypes.byref(slot)))
        return int(slot.value)

    def ring_wait(self, slot):
        self._chk(self._L.svo_ring_wait(self._h, int(slot)))

    def ring_query(self, slot):
        done, first, n, ms = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_float()
        self._chk(self._L.svo_ring_query(self._h, int(slot), ctypes.byref(done), ctypes.byref(first), ctypes.byref(n), ctypes.byref(ms)))
        return {"done": bool(done.value), "first_frame": int(first.value), "nframes": int(n.value), "gpu_ms": float(ms.value)}

    def ring_read(self, slot, k=0, want_hits=False):
        out = {"rgba": np.zeros((self.height, self.width, 4), dtype=np.uint8),
               "depth": np.zeros((self.height, self.width), dtype=np.float32)}
        self._chk(self._L.svo_ring_read_color(self._h, int(slot), int(k), out["rgba"].ctypes.data))
        self._chk(self._L.svo_ring_read_depth(self._h, int(slot), int(k), out["depth"].ctypes.data))
        if want_hits:
            out["hits"] = np.zeros((self.height, self.width), dtype=HIT_DTYPE)
            self._chk(self._L.svo_ring_read_hits(self._h, int(slot), int(k), out["hits"].ctypes.data))
        return out

    def ring_read_pixel(self, slot, k, x, y, want_hit=True):
        rgba = np.zeros(4, dtype=np.uint8)
        depth = np.zeros(1, dtype=np.float32)
        hit = np.zeros(1, dtype=HIT_DTYPE)
        self._chk(self._L.svo_ring_read_pixel(self._h, int(slot), int(k), int(x), int(y), rgba.ctypes.data, depth.ctypes.data,
                                              hit.ctypes.data if want_hit else None))
        return rgba, float(depth[0]), hit[0]

    def ring_bind_slot(self, slot, color_ptr, depth_ptr, hits_ptr, frame_stride):
        self._chk(self._L.svo_ring_bind_slot(self._h, int(slot), ctypes.c_void_p(color_ptr or 0), ctypes.c_void_p(depth_ptr or 0),
                                             ctypes.c_void_p(hits_ptr or 0), int(frame_stride)))

    def ring_forward_slot(self, slot, src, dst, nbytes, flag=None):
        self._chk(self._L.svo_ring_forward_slot(self._h, int(slot), ctypes.c_void_p(src or 0), ctypes.c_void_p(dst or 0), int(nbytes),
                                                ctypes.c_void_p(flag or 0)))

    # device memory shared between the ranks of a node
    def dev_alloc(self, nbytes):
        p = ctypes.c_void_p()
        self._chk(self._L.svo_dev_alloc(self._h, int(nbytes), ctypes.byref(p)))
        return p.value

    def dev_free(self, ptr):
        self._chk(self._L.svo_dev_free(self._h, ctypes.c_void_p(ptr)))

    def dev_read(self, ptr, nbytes, dtype=np.uint8):
        out = np.zeros(int(nbytes), dtype=np.uint8)
        self._chk(self._L.svo_dev_read(self._h, ctypes.c_void_p(ptr), out.ctypes.data, out.size))
        return out.view(dtype)

    def ipc_export(self, ptr):
        h = (ctypes.c_uint8 * 64)()
        self._chk(self._L.svo_ipc_export(self._h, ctypes.c_void_p(ptr), h))
        return bytes(h)

    def ipc_open(self, handle):
        buf = (ctypes.c_uint8 * 64).from_buffer_copy(handle)
        p = ctypes.c_void_p()
        self._chk(self._L.svo_ipc_open(self._h, buf, ctypes.byref(p)))
        return p.value

    def ipc_close(self, ptr):
        self._chk(self._L.svo_ipc_close(self._h, ctypes.c_void_p(ptr)))

    def ring_device_ptrs(self, slot):
        a, b, c, st = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        n = ctypes.c_uint64()
        self._chk(self._L.svo_ring_device_ptrs(self._h, int(slot), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c),
                                               ctypes.byref(n), ctypes.byref(st)))
        return {"color": a.value, "depth": b.value, "hits": c.value, "frame_stride": int(n.value), "stream": st.value}

    # dispatch
    def dispatch(self):
        self._chk(self._L.svo_dispatch(self._h))

    def dispatch_async(self):
        self._chk(self._L.svo_dispatch_async(self._h))

    def set_pick(self, x, y):
        """the pixel read_pixel answers without waiting for its frame (default: the image centre); x < 0: none"""
        self._chk(self._L.svo_set_pick(self._h, int(x), int(y)))

    def pick_info(self):
        x, y, m, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_uint64(), ctypes.c_uint64()
        self._chk(self._L.svo_pick_info(self._h, ctypes.byref(x), ctypes.byref(y), ctypes.byref(m), ctypes.byref(w)))
        return {"x": x.value, "y": y.value, "from_mail": int(m.value), "waited": int(w.value)}

    def set_overlap(self, sets):
        """image sets dispatch_async takes turns on: False / 0 = one (no alternation), True / 1 = the library's default, 2 .. 8 = that many"""
        self._chk(self._L.svo_set_overlap(self._h, int(sets)))

    def sync(self):
        self._chk(self._L.svo_sync(self._h))

    def count_frame(self):
        st = Stats()
        self._chk(self._L.svo_count_frame(self._h, ctypes.byref(st)))
        return st.as_dict()

    def stats(self):
        st = Stats()
        self._chk(self._L.svo_get_stats(self._h, ctypes.byref(st)))
        return st.as_dict()

    def time_frames(self, warmup, iters):
        ms = np.zeros(int(iters), dtype=np.float32)
        self._chk(self._L.svo_time_frames(self._h, int(warmup), int(iters),
                                          ms.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return ms

    # readback
    def read_color(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._chk(self._L.svo_read_color(self._h, out.ctypes.data))
        return out

    def read_depth(self):
        out = np.zeros((self.height, self.width), dtype=np.float32)
        self._chk(self._L.svo_read_depth(self._h, out.ctypes.data))
        return out

    def read_hits(self):
        out = np.zeros((self.height, self.width), dtype=HIT_DTYPE)
        self._chk(self._L.svo_read_hits(self._h, out.ctypes.data))
        return out

    def read_beam(self):
        out = np.zeros(((self.height + 3) // 4, (self.width + 3) // 4), dtype=np.float32)
        self._chk(self._L.svo_read_beam(self._h, out.ctypes.data))
        return out

    def read_pixel(self, x, y):
        rgba = np.zeros(4, dtype=np.uint8)
        depth = np.zeros(1, dtype=np.float32)
        hit = np.zeros(1, dtype=HIT_DTYPE)
        self._chk(self._L.svo_read_pixel(self._h, int(x), int(y), rgba.ctypes.data, depth.ctypes.data, hit.ctypes.data))
        return rgba, float(depth[0]), hit[0]

    def output_device_ptrs(self):
        a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._chk(self._L.svo_output_device_ptrs(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    # convenience used by tests / bench: one full frame, everything read back
    def render(self, pool=None, width=None, height=None, cam=None, frame_number=2, render_mode=2, bounces=2,
               mirror_mask=0, spp=1, use_beam=0):
        if pool is not None:
            self.pool_upload(pool)
        if width is not None:
            self.resize(width, height)
        if cam is not None:
            self.set_camera(cam)
        self.set_params(frame_number, render_mode, 0, use_beam, bounces, mirror_mask, spp)
        self.dispatch()
        return {"rgba": self.read_color(), "depth": self.read_depth(), "hits": self.read_hits()}
