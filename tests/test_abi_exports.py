"""The C-ABI libraries load on a CPU-only box and export every symbol their headers declare (no compute is called)."""
import ctypes
import os
import re

from srba_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    txt = open(os.path.join(ROOT, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_hip_abi_exports_every_declared_symbol():
    lib = capi.hip_lib()
    names = _declared("include/srba_hip.h", "srba_")
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), n


def test_engine_capi_exports_every_declared_symbol():
    lib = capi.engine_lib()
    for n in _declared("srba_amd/csrc/engine_capi.h", "srba_"):
        assert hasattr(lib, n), n


def test_struct_sizes_match_the_c_side():
    lib = capi.hip_lib()
    P, L, O, PD = (ctypes.c_int32(),) * 4
    P, L, O, PD = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    for fam, dims in capi.DIMS.items():
        assert lib.srba_family_dims(fam, ctypes.byref(P), ctypes.byref(L), ctypes.byref(O), ctypes.byref(PD)) == 0
        assert (P.value, L.value, O.value, PD.value) == dims
    p = capi.HipParams()
    lib.srba_hip_params_default(ctypes.byref(p), capi.SE3_STEREO)
    # reference defaults, include/srba/impl/rba_problem_common.h:35-56
    assert (p.max_iters, p.max_rho, p.max_lambda, p.max_error_per_obs_to_stop, p.min_error_reduction_ratio_to_relinearize, p.kernel_param) == (20, 10.0, 1e20, 1e-6, 0.01, 3.0)
    assert p.right_cam_pose[3] == 1.0 and p.sensor_pose_se3[3] == 1.0


def test_no_gpu_means_loud_failure():
    """The product path must not fall back to the CPU: without a device, context creation fails with a message."""
    import torch
    if torch.cuda.is_available():
        return
    lib = capi.hip_lib()
    p = capi.HipParams(); lib.srba_hip_params_default(ctypes.byref(p), capi.SE2_RELPOSE2D)
    assert not lib.srba_hip_create(-1, ctypes.byref(p))
    assert b"HIP" in lib.srba_hip_last_error(None) or b"device" in lib.srba_hip_last_error(None)
