"""raytrace_amd -- MI355X-native path-tracing core behind the s7a9/RayTrace interface.

The product is librtcore.so (HIP kernels + C ABI, see include/rtcore.h) and the
C++ host programs built on it.  The Python modules here are thin plumbing used
by the tests and the benchmark: `capi` binds the C ABI with ctypes, `scene`
assembles upstream-layout scene arrays, `procgen` generates stand-in models.
"""
__all__ = ["capi", "scene", "procgen"]
