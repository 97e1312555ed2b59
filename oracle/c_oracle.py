"""ctypes wrapper of oracle/liboracle.so (C restatement; TEST INFRASTRUCTURE ONLY -- see c_oracle.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


class Closure(C.Structure):
    _fields_ = [("kind", C.c_int32), ("flags", C.c_int32), ("n", C.c_int32), ("reserved", C.c_int32),
                ("coef", C.c_double * 16)]


def closure(kind=0, flags=0, coef=(0.0,)):
    c = Closure()
    c.kind, c.flags, c.n = kind, flags, len(coef)
    for i, v in enumerate(coef):
        c.coef[i] = float(v)
    return c


def load():
    if not os.path.exists(_LIB):
        subprocess.run(["make", "-s", "-C", _HERE], check=True)
    lib = C.CDLL(_LIB)
    lib.oracle_max_threads.restype = C.c_int
    return lib


def _suffix(a):
    return {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[a.dtype]


def rhs(eq, u, hx, hy, kappa, cmu, cmob):
    """eq: 0 Cahn-Hilliard, 1 Allen-Cahn; u is (nx, ny) float32/float64."""
    lib = load()
    u = np.ascontiguousarray(u)
    out, w1, w2 = np.empty_like(u), np.empty_like(u), np.empty_like(u)
    fn = getattr(lib, "oracle_rhs_" + _suffix(u))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    fn(C.c_int(eq), p(u), p(out), p(w1), p(w2), C.c_int(u.shape[0]), C.c_int(u.shape[1]), C.c_double(hx),
       C.c_double(hy), C.c_double(kappa), C.byref(cmu), C.byref(cmob))
    return out


def rk4(eq, y, hx, hy, kappa, cmu, cmob, dt, n, threads=None):
    lib = load()
    if threads:
        lib.oracle_set_threads(C.c_int(threads))
    y = np.array(y, copy=True, order="C")
    scratch = np.empty((5,) + y.shape, dtype=y.dtype)
    fn = getattr(lib, "oracle_rk4_" + _suffix(y))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    fn(C.c_int(eq), p(y), p(scratch), C.c_int(y.shape[0]), C.c_int(y.shape[1]), C.c_double(hx), C.c_double(hy),
       C.c_double(kappa), C.byref(cmu), C.byref(cmob), C.c_double(dt), C.c_int64(n))
    return y
