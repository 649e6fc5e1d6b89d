"""terra_amd -- MI355X-native core of the Terra path tracer's hot path.

The product is the shared library ``terra_amd/libterra_amd.so`` (HIP kernels +
C-ABI, sources in ``terra_amd/csrc``); this package is the thin host-side mirror
used by tests and the benchmark:

* ``terra_amd.api``      ctypes mirror of include/Terra.h / TerraPresets.h
* ``terra_amd.scenes``   synthetic scenes of BASELINE.json's configs
* ``terra_amd.runtime``  loading, device framebuffers, multi-GPU tile sharding
* ``terra_amd.build``    hipcc build of the library (gfx950)

There is no CPU fallback: `runtime.load()` raises if the library is missing.
"""
__all__ = ["api", "scenes", "runtime", "build"]
