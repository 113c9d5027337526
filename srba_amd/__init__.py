"""srba_amd -- MI355X-native implementation of SRBA's local-optimisation hot path.

The product is native code: the HIP kernels + C ABI in `srba_amd/csrc` (-> `lib/libsrba_hip.so`, declared in
`include/srba_hip.h`) and the header-only C++ front-end `include/srba.h` that keeps the reference's `srba::RbaEngine<>` API.
This Python package is only a ctypes driver used by `tests/` and `bench.py`.
"""
import os as _os
# the launch plan of libsrba_hip uses 16 concurrent streams; the HIP runtime reads this when it initialises (see srba_hip.hip)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from . import capi  # noqa: F401,E402

__all__ = ["capi", "datasets", "runner"]
