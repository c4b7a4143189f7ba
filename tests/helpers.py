"""Shared test helpers: golden-fixture access and comparison utilities."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "llvmpipe_golden.npz")
_cache = {}


def golden():
    if "z" not in _cache:
        _cache["z"] = np.load(GOLDEN)
    return _cache["z"]


def golden_cases():
    z = golden()
    return [tuple(s.split(":")) for s in z["index"].tolist()]


def golden_case(name, poolkey):
    z = golden()
    w, h, frame, mode, same = [int(v) for v in z[name + "/meta"]]
    return {
        "pool": z["pool/" + poolkey], "w": w, "h": h, "frame": frame, "mode": mode,
        "cam": z[name + "/cam"], "rgba": z[name + "/rgba"], "depth_bits": z[name + "/depth_bits"],
        "first_hit": z[name + "/first_hit"], "patched_same": bool(same),
    }


def nan_colour_mask(hits, mode):
    """Pixels whose radiance is NaN-derived in the reference: a
    primary hit whose packed normal decodes to the zero vector."""
    hit = hits["pointer"] != 0
    rn = hits["raw_normal"]
    if mode == 0:
        return hit & ((rn == 0) | (rn == 555))
    if mode in (2, 3):
        return hit & (rn == 555)
    return np.zeros_like(hit)


def compare_with_golden(res, g):
    """res: dict(rgba, depth, hits) from any implementation. Returns dict of mismatch counts."""
    fh = g["first_hit"]
    hits = res["hits"]
    hit = fh[..., 0] != 0
    out = {
        "rgba": int((res["rgba"] != g["rgba"]).any(axis=2).sum()),
        "depth": int((res["depth"].view(np.uint32) != g["depth_bits"]).sum()),
        "pointer": int((hits["pointer"] != fh[..., 0]).sum()),
        "value": int(((hits["value"] != fh[..., 1]) & hit).sum()),
        "raw_normal": int(((hits["raw_normal"] != fh[..., 2]) & hit).sum()),
        "level": int(((hits["level"] != (fh[..., 3] >> 16)) & hit).sum()),
        "iter": int(((hits["iter"] != (fh[..., 3] & 0xFFFF)) & hit).sum()),
    }
    return out


class DualContext:
    """A test's renderer over BOTH libraries: libsvohip.so (what a host loads: pipelines 0 and 1) and libsvohip_variants.so (the
    comparators: pipeline 2, the spare-ray kernel, the A/B environment switches).  set_pipeline(2) moves the test onto a context
    of the variants library, set_pipeline(0 / 1) back onto the product library's; every other call goes to the context in force.
    What a test set up before it switched -- pool, image size, camera, frame parameters, ... -- follows: the last call of each
    such setter is journalled and replayed, in order, on the context that comes into force.  Everything pipelines 0 and 1 are
    tested on therefore runs on the library that ships."""
    STICKY = ("pool_upload", "build_from_heightmap", "build_from_heightmap16", "build_from_voxels", "resize", "set_camera", "set_params",
              "set_hit_records", "set_tuning", "set_progressive", "set_sequence", "set_batch", "set_rows", "set_stripes", "set_derived",
              "set_pick", "set_overlap", "set_reserved_cus")

    def __init__(self, device=0, variants=False):
        from svo_raytracer_amd import hiplib
        self._hiplib, self._device = hiplib, device
        self._ctx = {False: None, True: None}
        self._journal = []                       # [(serial, name, args, kwargs)], the last call of each setter, oldest first
        self._applied = {False: 0, True: 0}      # serial up to which each context has seen the journal
        self._serial = 0
        self._variants = bool(variants)
        self._get(self._variants)

    def _get(self, variants):
        if self._ctx[variants] is None:
            path = self._hiplib.VARIANTS_LIB_PATH if variants else None
            if variants and not os.path.exists(path):
                raise RuntimeError("libsvohip_variants.so missing: run __graft_entry__.build()")
            self._ctx[variants] = self._hiplib.HipContext(self._device, lib_path=path)
        c = self._ctx[variants]
        for serial, name, args, kwargs in self._journal:
            if serial > self._applied[variants]:
                getattr(c, name)(*args, **kwargs)
        self._applied[variants] = self._serial
        return c

    @property
    def active(self):
        return self._ctx[self._variants]

    def set_pipeline(self, p):
        self._variants = int(p) == 2
        self._get(self._variants).set_pipeline(p)

    def use_variants(self, on=True):
        """the variants library whatever the pipeline (tests of the spare-ray kernel and of the environment switches)"""
        self._variants = bool(on)
        return self._get(self._variants)

    def __getattr__(self, name):
        target = getattr(self._ctx[self._variants], name)
        if name not in self.STICKY:
            return target

        def call(*args, **kwargs):
            out = target(*args, **kwargs)
            self._serial += 1
            if name in ("build_from_heightmap", "build_from_heightmap16", "build_from_voxels", "pool_upload"):   # the latest pool wins
                self._journal = [j for j in self._journal if j[1] not in ("build_from_heightmap", "build_from_heightmap16", "build_from_voxels", "pool_upload")]
            self._journal = [j for j in self._journal if j[1] != name] + [(self._serial, name, args, kwargs)]
            self._applied[self._variants] = self._serial
            return out
        return call

    def render(self, pool=None, width=None, height=None, cam=None, frame_number=2, render_mode=2, bounces=2, mirror_mask=0, spp=1, use_beam=0):
        # (HipContext.render sets state through its own methods: journal the same calls here)
        if pool is not None:
            self.pool_upload(pool)
        if width is not None:
            self.resize(width, height)
        if cam is not None:
            self.set_camera(cam)
        self.set_params(frame_number, render_mode, 0, use_beam, bounces, mirror_mask, spp)
        a = self.active
        a.dispatch()
        return {"rgba": a.read_color(), "depth": a.read_depth(), "hits": a.read_hits()}

    def close(self):
        for k in (False, True):
            if self._ctx[k] is not None:
                self._ctx[k].close()
                self._ctx[k] = None
