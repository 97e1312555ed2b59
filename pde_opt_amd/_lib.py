"""ctypes binding of libpdeopt_hip.so (the C ABI declared in include/pdeopt_hip.h).

There is NO CPU fallback: if the shared library is missing or no HIP device is present the
constructors below raise ``HipUnavailableError``.  Build the library with
``python -m pde_opt_amd.csrc.build`` (or ``__graft_entry__.build()``).
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDEOPT_LIB") or os.path.join(_HERE, "libpdeopt_hip.so")  # PDEOPT_LIB: A/B builds

MAX_COEF = 16

# enums (mirror include/pdeopt_hip.h)
OK, EINVAL, EHIP, EFFT, ENONFINITE, ENOMEM, ESTATE = range(7)
F32, F64 = 0, 1
DERIVS_FD, DERIVS_FOURIER = 0, 1
EQ_CAHN_HILLIARD, EQ_ALLEN_CAHN, EQ_ADVECTION_DIFFUSION, EQ_GPE = 0, 1, 2, 3
EQ_ALLEN_CAHN_SBM, EQ_CAHN_HILLIARD_SBM = 4, 5
EQ_CAHN_HILLIARD_3D = 6
EQ_SHAPE_SMOOTH = 7  # Shape.smooth_shape (shapes.py:39-64)
INT_EULER, INT_RK4, INT_IMEX, INT_STRANG, INT_TSIT5 = 0, 1, 2, 3, 4
CL_POLY, CL_LEGENDRE, CL_JIT = 0, 1, 2
CL_LOGIT_PRIOR, CL_EXP_WRAP = 1, 2
AUX_VX_FACE, AUX_VY_FACE, AUX_IMEX_SYMBOL, AUX_GPE_A_TERM, AUX_GPE_POTENTIAL = 0, 1, 2, 3, 4
AUX_SBM_PSI, AUX_SBM_NORM_GRAD, AUX_SBM_MASK = 5, 6, 7
RED_MEAN, RED_VAR, RED_MIN, RED_MAX, RED_SUMSQ, RED_NONFINITE = 0, 1, 2, 3, 4, 5
OPT_KERNEL_PATH = 0
OPT_TILE_ROWS = 1
OPT_GROUP_ENVS = 2
OPT_DEBUG_ABLATE = 3
OPT_FUSE_STAGES = 4
OPT_HALO_LAYOUT = 5
OPT_GRAPH = 6
OPT_IMEX_LDS_FFT = 7
OPT_SMALL_PERSIST = 8
OPT_GROUP_STREAMS = 9
CNT_STAGE_LAUNCHES = 0
CNT_LAST_GROUPS = 1
CNT_GROUP_STREAMS = 2
COPY_H2D, COPY_D2H, COPY_D2D = 0, 1, 2
FIELD_Y, FIELD_TA, FIELD_TB, FIELD_ACC = 0, 1, 2, 3
PATH_AUTO, PATH_GENERIC, PATH_TILED = 0, 1, 2
PART_ALL, PART_INTERIOR, PART_EDGE = 0, 1, 2


class HipUnavailableError(RuntimeError):
    """The HIP library or a GPU is missing; the product path has no CPU fallback."""


class PdeoptError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pdeopt status {code}: {msg}")
        self.code = code


class Closure(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("flags", C.c_int32),
        ("n", C.c_int32),
        ("reserved", C.c_int32),
        ("coef", C.c_double * MAX_COEF),
    ]


MAX_SPOTS = 4


class LightSpot(C.Structure):  # pdeopt_light_spot
    _fields_ = [(n, C.c_double) for n in ("amp0", "amp_rate", "x0", "x_rate", "y0", "y_rate", "inv_two_w2")]


class Problem(C.Structure):
    _fields_ = [
        ("equation", C.c_int32),
        ("dtype", C.c_int32),
        ("nx", C.c_int32),
        ("ny", C.c_int32),
        ("batch", C.c_int32),
        ("derivs", C.c_int32),
        ("hx", C.c_double),
        ("hy", C.c_double),
        ("kappa", C.c_double),
        ("mu", Closure),
        ("mob", Closure),
        ("gpe_k", C.c_double),
        ("fe", Closure),
        ("nz", C.c_int32),
        ("reserved2", C.c_int32),
        ("hz", C.c_double),
    ]


# every symbol include/pdeopt_hip.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
# pdeopt_time_fn: void (*)(double t, double out[3], void* user)
TIME_FN = C.CFUNCTYPE(None, C.c_double, C.POINTER(C.c_double), C.c_void_p)
# pdeopt_aux_fn: int (*)(double t, int which, void* host_out, void* user)
AUX_FN = C.CFUNCTYPE(C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p)
class Pid(C.Structure):
    """pdeopt_pid (include/pdeopt_hip.h)"""
    _fields_ = [(n, C.c_double) for n in ("rtol", "atol", "pcoeff", "icoeff", "dcoeff", "dtmin", "dtmax", "factormin", "factormax", "safety")]


class Tsit5Stats(C.Structure):
    """pdeopt_tsit5_stats (include/pdeopt_hip.h)"""
    _fields_ = [("t", C.c_double), ("dt", C.c_double), ("accepted", C.c_int64), ("rejected", C.c_int64), ("status", C.c_int32),
                ("saved", C.c_int32)]


TSIT5_DONE, TSIT5_MAX_STEPS, TSIT5_STALLED = 0, 1, 2

_SIGNATURES = {
    "pdeopt_abi_version": (C.c_int, []),
    "pdeopt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pdeopt_ctx_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "pdeopt_ctx_destroy": (C.c_int, [_VP]),
    "pdeopt_last_error": (C.c_char_p, [_VP]),
    "pdeopt_set_option": (C.c_int, [_VP, C.c_int, C.c_int64]),
    "pdeopt_configure": (C.c_int, [_VP, C.POINTER(Problem)]),
    "pdeopt_set_env_params": (C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP, _VP]),
    "pdeopt_set_aux": (C.c_int, [_VP, C.c_int, _VP, C.c_int]),
    "pdeopt_set_aux_time_fn": (C.c_int, [_VP, C.c_int, AUX_FN, _VP, C.c_int]),
    "pdeopt_set_env_gpe_k": (C.c_int, [_VP, C.c_int, C.c_int, _VP]),
    "pdeopt_set_env_imex_scale": (C.c_int, [_VP, C.c_int, C.c_int, _VP]),
    "pdeopt_set_gpe_spots": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, _VP, C.c_double, C.c_double]),
    "pdeopt_set_state": (C.c_int, [_VP, C.c_int, C.c_int, _VP]),
    "pdeopt_get_state": (C.c_int, [_VP, C.c_int, C.c_int, _VP]),
    "pdeopt_state_device_ptr": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_int64)]),
    "pdeopt_rhs": (C.c_int, [_VP, C.c_double, _VP]),
    "pdeopt_advance": (C.c_int, [_VP, C.c_int, C.c_double, C.c_double, C.c_int64]),
    "pdeopt_set_integrator_params": (C.c_int, [_VP, C.c_double, C.c_double, C.c_double, C.c_double]),
    "pdeopt_set_jit_closures": (C.c_int, [_VP, C.c_char_p, C.c_char_p]),
    "pdeopt_jit_check": (C.c_int, [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
    "pdeopt_set_time_terms": (C.c_int, [_VP, TIME_FN, _VP, C.POINTER(C.c_double)]),
    "pdeopt_set_time_table": (C.c_int, [_VP, C.c_int, _VP, _VP]),
    "pdeopt_set_time_terms_poly": (C.c_int, [_VP, C.c_int, _VP, C.c_int, _VP]),
    "pdeopt_snapshot": (C.c_int, [_VP]),
    "pdeopt_get_interpolated": (C.c_int, [_VP, C.c_double, C.c_int, C.c_int, _VP]),
    "pdeopt_reduce": (C.c_int, [_VP, C.c_int, _VP]),
    "pdeopt_probe": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, _VP]),
    "pdeopt_observe_u8": (C.c_int, [_VP, C.c_double, C.c_double, C.c_int, C.c_int, _VP]),
    "pdeopt_observe_u8_device": (C.c_int, [_VP, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "pdeopt_detect_vortices": (C.c_int, [_VP, C.c_double, C.c_double, C.c_int, C.c_int, _VP, _VP]),
    "pdeopt_tsit5_trial": (C.c_int, [_VP, C.c_double, C.c_double, C.c_double, C.c_double, _VP]),
    "pdeopt_tsit5_commit": (C.c_int, [_VP, C.c_int]),
    "pdeopt_tsit5_trial_env": (C.c_int, [_VP, C.c_double, _VP, C.c_double, C.c_double, C.POINTER(C.c_double), _VP]),
    "pdeopt_tsit5_commit_env": (C.c_int, [_VP, _VP]),
    "pdeopt_tsit5_dense": (C.c_int, [_VP, C.c_double, C.c_double, C.c_int, C.c_int, _VP]),
    "pdeopt_tsit5_solve_small_supported": (C.c_int, [_VP]),
    "pdeopt_tsit5_solve_small": (C.c_int, [_VP, C.c_double, C.c_double, C.c_double, _VP, C.c_int64, C.c_int, _VP, _VP, _VP]),
    "pdeopt_halo_strip_elems": (C.c_int, [_VP, C.POINTER(C.c_int64)]),
    "pdeopt_halo_pack": (C.c_int, [_VP, C.c_int, _VP]),
    "pdeopt_halo_unpack": (C.c_int, [_VP, C.c_int, _VP, C.POINTER(C.c_int)]),
    "pdeopt_rk4_phase_plan": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pdeopt_rk4_phase": (C.c_int, [_VP, C.c_int, C.c_double]),
    "pdeopt_rk4_phase_part": (C.c_int, [_VP, C.c_int, C.c_double, C.c_int]),
    "pdeopt_rk4_loopback_advance": (C.c_int, [_VP, C.c_double, C.c_int64]),
    "pdeopt_comm_unique_id": (C.c_int, [C.c_char_p]),
    "pdeopt_comm_init": (C.c_int, [_VP, C.c_int, C.c_int, C.c_char_p]),
    "pdeopt_comm_destroy": (C.c_int, [_VP]),
    "pdeopt_local_group_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "pdeopt_local_group_destroy": (C.c_int, [_VP]),
    "pdeopt_comm_init_local": (C.c_int, [_VP, _VP, C.c_int]),
    "pdeopt_comm_ipc_export": (C.c_int, [_VP, C.c_int, C.c_int, _VP]),
    "pdeopt_comm_ipc_attach": (C.c_int, [_VP, _VP]),
    "pdeopt_rk4_decomposed_advance": (C.c_int, [_VP, C.c_double, C.c_int64, C.POINTER(C.c_int), C.c_int]),
    "pdeopt_ctx_create_on_stream": (C.c_int, [C.c_int, _VP, C.POINTER(_VP)]),
    "pdeopt_host_alloc": (C.c_int, [_VP, C.c_int64, C.POINTER(_VP)]),
    "pdeopt_host_free": (C.c_int, [_VP, _VP]),
    "pdeopt_buffer_alloc": (C.c_int, [_VP, C.c_int64, C.POINTER(_VP)]),
    "pdeopt_buffer_free": (C.c_int, [_VP, _VP]),
    "pdeopt_buffer_copy": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int]),
    "pdeopt_get_counter": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_int64)]),
    "pdeopt_sync": (C.c_int, [_VP]),
    "pdeopt_timer_start": (C.c_int, [_VP]),
    "pdeopt_timer_stop": (C.c_int, [_VP, C.POINTER(C.c_double)]),
    "pdeopt_timer_clock": (C.c_int, [_VP, C.POINTER(C.c_double)]),
    "pdeopt_last_kernel": (C.c_char_p, [_VP]),
}

_lib = None


def load_library():
    """dlopen libpdeopt_hip.so and bind every declared symbol; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipUnavailableError(
            f"{LIB_PATH} not found. pde_opt_amd has no CPU fallback: build the HIP library with "
            "`python -m pde_opt_amd.csrc.build` (needs hipcc, targets gfx950)."
        )
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # missing ROCm runtime etc.
        raise HipUnavailableError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch: let it propagate
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def device_count() -> int:
    lib = load_library()
    n = C.c_int(0)
    lib.pdeopt_device_count(C.byref(n))
    return n.value


def np_dtype(code: int):
    return np.float32 if code == F32 else np.float64


def dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    raise ValueError(f"unsupported dtype {dt}; the HIP path computes in float32 or float64")
