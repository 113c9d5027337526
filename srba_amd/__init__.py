"""srba_amd -- MI355X-native implementation of SRBA's local-optimisation hot path.

The product is native code: the HIP kernels + C ABI in `srba_amd/csrc` (-> `lib/libsrba_hip.so`, declared in
`include/srba_hip.h`) and the header-only C++ front-end `include/srba.h` that keeps the reference's `srba::RbaEngine<>` API.
This Python package is only a ctypes driver used by `tests/` and `bench.py`.
"""
from . import capi  # noqa: F401

__all__ = ["capi", "datasets", "runner"]
