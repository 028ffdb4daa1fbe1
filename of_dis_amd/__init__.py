"""of_dis_amd -- MI355X-native implementation of the OF_DIS hot path.

The product is the C-ABI shared library `of_dis_amd/lib/libofdis_hip.so` (HIP kernels for gfx950 +
C++ host code, include/ofdis.h) and the `run_OF_INT` / `run_OF_RGB` executables.  This Python
package is only a ctypes binding of that ABI for the test-suite and the benchmark harness.
Nothing here falls back to a CPU implementation: using the binding without the built library
raises.
"""
from .params import OfdisParams, oppoint, padded_size, auto_first_scale  # noqa: F401

__all__ = ["OfdisParams", "oppoint", "padded_size", "auto_first_scale", "capi"]
