"""svo-raytracer_amd: MI355X-native SVO ray traversal + path tracing hot path.

Only what the hot path needs lives here:
  csrc/    HIP kernels (gfx950) + the C-ABI shared library (include/svo_hip.h)
  host/    C++ host-side mirror of the reference's Renderer / Camera / Octree
  scene/   deterministic procedural SVO scene generator (reference pool layout)
  *.py     thin ctypes bindings used by tests/ and bench.py
"""
__all__ = ["scene", "hiplib", "hostlib", "cameras", "tiles"]
