"""lidar_imu_init_amd — MI355X-native (gfx950) hot path of LI-Init behind a C-ABI.

The product is `lib/libliinit_hip.so` (hand-written HIP kernels + C++ host, see include/liinit_hip.h).
This package is the thin Python mirror used by tests, bench.py and harnesses: ctypes bindings only —
no compute happens in Python and nothing here falls back to a CPU implementation.
"""
from .api import (LIIError, Registrar, State, calib_state_array, library_path, load_library,  # noqa: F401
                  pose6d_array)

__all__ = ["LIIError", "Registrar", "State", "load_library", "library_path", "calib_state_array", "pose6d_array"]
