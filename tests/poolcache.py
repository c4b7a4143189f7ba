"""The 8192^3 pools several test modules need, built once per session (a build is 20 s - 100 s of the host's cores; a pool is
1.4 - 2.0 GB, so at most two are kept)."""
import collections

import svo_raytracer_amd.scene as scene

_cache = collections.OrderedDict()


def pool(family="terrain", n=8192, seed=1, amp=8, dens=scene.CAVES_DENS):
    key = (family, n, seed, amp, dens if family != "terrain" else 0)
    if key in _cache:
        _cache.move_to_end(key)
        return _cache[key]
    p, _ = scene.build(family, n, seed, amp, dens, dens)      # (dens: the caves' density or the dust's, by family)
    p.setflags(write=False)
    _cache[key] = p
    while len(_cache) > 2:
        _cache.popitem(last=False)
    return p
