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
