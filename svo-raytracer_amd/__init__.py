"""svo-raytracer_amd: MI355X-native SVO ray traversal + path tracing hot path.

Only what the hot path needs lives here:
  csrc/    HIP kernels (gfx950) + the C-ABI shared library (include/svo_hip.h)
  host/    C++ host-side mirror of the reference's Renderer / Camera / Octree
  scene/   deterministic procedural SVO scene generator (reference pool layout)
  *.py     thin ctypes bindings used by tests/ and bench.py; framering.py / tiles.py: frames in flight and the
           multi-GPU tile-stripe split + gather that bench.py drives
"""
__all__ = ["scene", "hiplib", "hostlib", "cameras", "tiles", "framering"]


def one_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64; libsvohip.so links the system ROCm by soname.
    Two HIP runtimes in one process do not share the GPU (the second finds no device).  Loading torch's libraries
    first makes the dynamic linker resolve libsvohip.so's dependency to the runtime that is already mapped, so a
    Python process that uses both (bench.py, the tests: torch owns the gather buffers and RCCL) has exactly one.
    Importing torch does not touch the GPU.  A process without torch (the Java host) simply uses the system ROCm."""
    try:
        import torch  # noqa: F401
    except Exception:
        pass
