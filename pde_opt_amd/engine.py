"""HipEngine: one pdeopt_ctx (one GPU, one HIP stream) behind a small Python object.

This is the only module that calls into libpdeopt_hip.so.  Everything the reference does between
``equation_type(domain=..., **params)`` and ``solution.ys[-1]`` in ``PDEEnv.step``
(pde_opt/pde_env.py:286-305) maps onto: ``configure`` -> ``set_state`` -> ``advance`` ->
``get_state``.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib as L
from .numerics.closures import ClosureDesc


def _closure_struct(desc: Optional[ClosureDesc]) -> L.Closure:
    s = L.Closure()
    if desc is None:
        s.kind, s.flags, s.n = L.CL_POLY, 0, 1
        s.coef[0] = 0.0
        return s
    s.kind, s.flags, s.n = desc.kind, desc.flags, len(desc.coef)
    for i, v in enumerate(desc.coef):
        s.coef[i] = float(v)
    return s


class HipEngine:
    """Owns a device context and the field buffers of one batched problem."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        """``stream``: optional HIP stream handle (e.g. ``torch.cuda.current_stream().cuda_stream``)
        to order this engine's work on a caller-owned stream."""
        self._lib = L.load_library()
        n = L.device_count()
        if n <= 0:
            raise L.HipUnavailableError(
                "no HIP device visible: pde_opt_amd runs its hot path on an MI355X only "
                "(there is no CPU fallback)"
            )
        h = C.c_void_p()
        if stream:
            rc = self._lib.pdeopt_ctx_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(h))
        else:
            rc = self._lib.pdeopt_ctx_create(int(device), C.byref(h))
        if rc != L.OK:
            raise L.PdeoptError(rc, self._lib.pdeopt_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self.problem: Optional[L.Problem] = None
        self._key = None
        self.dtype = np.float32
        self.batch = 0
        self.state_shape: tuple = ()
        self._aux_thunks: dict = {}
        self._aux_keys: dict = {}
        self._callback_error = None

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc: int):
        err, self._callback_error = getattr(self, "_callback_error", None), None
        if err is not None:  # a Python callback raised inside the library call: re-raise it here
            raise err
        if rc != L.OK:
            msg = self._lib.pdeopt_last_error(self._h).decode()
            if rc == L.EINVAL:
                raise ValueError(msg)
            raise L.PdeoptError(rc, msg)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pdeopt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def last_kernel(self) -> str:
        return self._lib.pdeopt_last_kernel(self._h).decode()

    def set_kernel_path(self, path: int):
        self._check(self._lib.pdeopt_set_option(self._h, L.OPT_KERNEL_PATH, int(path)))

    def set_tile_rows(self, rows: int):
        self._check(self._lib.pdeopt_set_option(self._h, L.OPT_TILE_ROWS, int(rows)))

    def set_graph(self, v: int):
        """hipGraph replay of the substep loop: 0 auto, 1 always, -1 never"""
        self._check(self._lib.pdeopt_set_option(self._h, L.OPT_GRAPH, int(v)))

    def set_fuse_stages(self, v: int):
        self._check(self._lib.pdeopt_set_option(self._h, L.OPT_FUSE_STAGES, int(v)))

    def set_group_streams(self, v: int):
        """explicit integrators running the batch in cache-resident groups: 0 auto (two groups side by side on two
        HIP streams), 1 one group at a time, 2 force two (``PDEOPT_OPT_GROUP_STREAMS``)"""
        self._check(self._lib.pdeopt_set_option(self._h, L.OPT_GROUP_STREAMS, int(v)))

    def set_small_persist(self, v: int):
        """whole-environment-step kernel for LDS-resident grids (Euler / RK4): 0 auto, 1 wherever it can run, -1 never"""
        self._check(self._lib.pdeopt_set_option(self._h, L.OPT_SMALL_PERSIST, int(v)))

    def set_group_envs(self, n: int):
        self._check(self._lib.pdeopt_set_option(self._h, L.OPT_GROUP_ENVS, int(n)))

    # -- problem ----------------------------------------------------------------------------
    def configure(
        self,
        equation: int,
        dtype,
        nx: int,
        ny: int,
        batch: int,
        hx: float,
        hy: float,
        kappa: float = 0.0,
        mu: Optional[ClosureDesc] = None,
        mob: Optional[ClosureDesc] = None,
        gpe_k: float = 0.0,
        derivs: int = 0,
        fe: Optional[ClosureDesc] = None,
        nz: int = 0,
        hz: float = 0.0,
    ):
        p = L.Problem()
        p.nz, p.hz = int(nz), float(hz)
        p.derivs = int(derivs)
        p.fe = _closure_struct(fe)
        p.equation, p.dtype = int(equation), L.dtype_code(dtype)
        p.nx, p.ny, p.batch = int(nx), int(ny), int(batch)
        p.hx, p.hy, p.kappa, p.gpe_k = float(hx), float(hy), float(kappa), float(gpe_k)
        p.mu, p.mob = _closure_struct(mu), _closure_struct(mob)
        # a closure outside the in-kernel family (kind JIT): both roles travel as C function bodies and the generic
        # stage kernel is compiled with them at run time (csrc/jit.hip)
        if any(d is not None and d.kind == L.CL_JIT for d in (mu, mob)):
            from .numerics.closures import jit_body_of

            if mu is None or mob is None:
                raise ValueError("run-time-compiled closures need both mu and the mobility")
            p.mu.kind = p.mob.kind = L.CL_JIT
            self._check(self._lib.pdeopt_set_jit_closures(self._h, jit_body_of(mu).encode(), jit_body_of(mob).encode()))
        old = self.problem
        if old is None or (old.equation, old.dtype, old.nx, old.ny, old.nz, old.batch) != (p.equation, p.dtype, p.nx, p.ny, p.nz, p.batch):
            self._aux_keys.clear()  # a new shape frees the library's auxiliary fields
        try:
            self._check(self._lib.pdeopt_configure(self._h, C.byref(p)))
        except Exception:
            # a refused configure leaves the ctx unconfigured: the NEXT configure frees every auxiliary field in
            # the library, whatever its shape -- forget what was uploaded so that it is transferred again
            self._aux_keys.clear()
            self._aux_thunks.clear()
            self.problem = None
            raise
        self.problem = p
        self.dtype = L.np_dtype(p.dtype)
        self.batch = int(batch)
        comps = (2,) if equation == L.EQ_GPE else ((int(nz),) if equation == L.EQ_CAHN_HILLIARD_3D else ())
        self.state_shape = (int(nx), int(ny)) + comps

    def set_env_params(self, env_first: int, kappa=None, mu_coef=None, mob_coef=None):
        """Per-environment control parameters (kappa and closure coefficient VALUES)."""
        if self.problem is not None and L.CL_JIT in (self.problem.mu.kind, self.problem.mob.kind):
            # run-time-compiled closures carry their constants in the compiled body: no per-environment coefficients
            for arr in (mu_coef, mob_coef):
                if arr is not None:
                    a = np.atleast_2d(np.asarray(arr, dtype=np.float64))
                    if np.any(a != a[0]):
                        raise ValueError("closures compiled at run time (outside the in-kernel family) cannot differ between the "
                                         "environments of one batch: per-environment closure coefficients need a family member")
        count = None
        bufs = []
        for arr, width in ((kappa, None), (mu_coef, L.MAX_COEF), (mob_coef, L.MAX_COEF)):
            if arr is None:
                bufs.append(None)
                continue
            a = np.asarray(arr, dtype=np.float64)
            if width is not None:
                a = np.atleast_2d(a)
                pad = np.zeros((a.shape[0], width))
                pad[:, : a.shape[1]] = a
                a = pad
            else:
                a = np.atleast_1d(a)
            a = np.ascontiguousarray(a)
            count = a.shape[0] if count is None else count
            if a.shape[0] != count:
                raise ValueError("per-environment parameter arrays disagree on the number of environments")
            bufs.append(a)
        if count is None:
            return
        ptrs = [b.ctypes.data_as(C.c_void_p) if b is not None else None for b in bufs]
        self._check(self._lib.pdeopt_set_env_params(self._h, int(env_first), int(count), *ptrs))

    def _aux_array(self, which: int, field, per_env: bool) -> np.ndarray:
        cplx = which in (L.AUX_IMEX_SYMBOL, L.AUX_GPE_A_TERM)
        p = self.problem
        if cplx:
            a = np.asarray(field, dtype=np.complex128 if self.dtype == np.float64 else np.complex64)
        else:
            a = np.asarray(field, dtype=self.dtype)
        want = ((p.batch,) if per_env else ()) + (p.nx, p.ny) + ((p.nz,) if p.nz > 1 else ())
        if a.shape != want:
            a = np.broadcast_to(a, want)
        return np.ascontiguousarray(a)

    def set_aux(self, which: int, field, per_env: bool = False, key=None):
        """``key``: identity of the field's contents (``KeyedArray.key``); a field this engine already holds under
        the same key is not transferred again"""
        which = int(which)
        if key is not None and self._aux_keys.get(which) == (key, bool(per_env)):
            return
        self._aux_keys.pop(which, None)
        a = self._aux_array(which, field, per_env)
        self._check(self._lib.pdeopt_set_aux(self._h, int(which), a.ctypes.data_as(C.c_void_p), int(per_env)))
        self._aux_thunks.pop(int(which), None)  # the static upload replaced a time-dependent source
        if key is not None:
            self._aux_keys[which] = (key, bool(per_env))

    def set_aux_time_fn(self, which: int, fn, per_env: bool = False):
        """Time-dependent auxiliary field: ``fn(t)`` returns the field ((nx, ny), or (batch, nx, ny) with
        ``per_env``) at local time ``t``; the library calls it once per distinct evaluation time from inside
        ``advance`` / ``rhs`` (every Strang substep's t0, every Runge-Kutta stage time), as the reference's
        ``terms.vf(t, y, args)`` does (numerics/solvers.py:109).  ``fn=None`` removes it."""
        which = int(which)
        if fn is None:
            self._check(self._lib.pdeopt_set_aux_time_fn(self._h, which, L.AUX_FN(0), None, 0))
            self._aux_thunks.pop(which, None)
            return

        def thunk(t, w, out_ptr, _user):
            try:
                a = self._aux_array(w, fn(t), per_env)
                C.memmove(out_ptr, a.ctypes.data, a.nbytes)
                return 0
            except BaseException as e:  # noqa: BLE001 -- must not propagate through the C frame
                self._callback_error = e
                return 1

        self._aux_keys.pop(which, None)
        cb = L.AUX_FN(thunk)
        self._check(self._lib.pdeopt_set_aux_time_fn(self._h, which, cb, None, int(per_env)))
        self._aux_thunks[which] = cb  # keep the ctypes thunk alive while the library may call it

    def set_gpe_spots(self, tables, x_first: float = 0.0, y_first: float = 0.0, env_first: int = 0):
        """Gaussian light spots of the GPE control field, evaluated in-kernel at every substep's t0.
        ``tables``: (envs, n_spots, 7) rows of ``pdeopt_light_spot`` (``GaussianSpots.table``); ``None`` or an
        empty second axis removes the spots of the whole batch."""
        if tables is None:
            self._check(self._lib.pdeopt_set_gpe_spots(self._h, 0, self.batch, 0, None, 0.0, 0.0))
            return
        a = np.ascontiguousarray(np.asarray(tables, dtype=np.float64))
        if a.ndim != 3 or a.shape[2] != 7:
            raise ValueError(f"spot tables have shape (envs, n_spots, 7), got {a.shape}")
        self._check(self._lib.pdeopt_set_gpe_spots(self._h, int(env_first), a.shape[0], a.shape[1],
                                                   a.ctypes.data_as(C.c_void_p), float(x_first), float(y_first)))

    def set_env_imex_scale(self, env_first: int, sigma):
        """IMEX: environment b integrates with ``sigma[b] x`` the uploaded ``fourier_symbol`` (a per-environment
        ``kappa``); all ones restores the paired transforms"""
        a = np.ascontiguousarray(np.atleast_1d(np.asarray(sigma, dtype=np.float64)))
        self._check(self._lib.pdeopt_set_env_imex_scale(self._h, int(env_first), a.shape[0], a.ctypes.data_as(C.c_void_p)))

    def set_env_gpe_k(self, env_first: int, k):
        """per-environment GPE interaction strength (the control value travels with the environment)"""
        a = np.ascontiguousarray(np.atleast_1d(np.asarray(k, dtype=np.float64)))
        self._check(self._lib.pdeopt_set_env_gpe_k(self._h, int(env_first), a.shape[0], a.ctypes.data_as(C.c_void_p)))

    def set_integrator_params(self, imex_A=0.5, time_scale=1.0, strang_dx=1.0):
        ts = complex(time_scale)
        self._check(
            self._lib.pdeopt_set_integrator_params(self._h, float(imex_A), ts.real, ts.imag, float(strang_dx))
        )

    def set_time_terms(self, fn=None, constant=(0.0, 0.0, 0.0), theta_poly=None, flux_poly=None):
        """Smoothed-boundary scalars per RHS evaluation time: ``fn(t) -> (cos theta on the mask,
        cos off the mask, flux)`` is called by the library at every stage; ``fn=None`` uses
        ``constant``.  The ctypes thunk is kept alive on the engine.  ``theta_poly`` / ``flux_poly``: ascending
        coefficients of theta(t) and flux(t) when both are polynomials of degree <= 3 (``closures.poly_in_t``) --
        the in-kernel adaptive solve evaluates them at its own stage times (``pdeopt_set_time_terms_poly``); without
        them a callback keeps an adaptive solve on the host-driven path."""
        const = (C.c_double * 3)(*[float(v) for v in constant])
        self._time_terms_fn = fn
        if fn is None:
            self._time_thunk = None
            self._check(self._lib.pdeopt_set_time_terms(self._h, L.TIME_FN(0), None, const))
            return
        self._set_time_thunk(fn, const)
        if theta_poly is not None and flux_poly is not None:
            th = np.ascontiguousarray(np.asarray(theta_poly, dtype=np.float64))
            fl = np.ascontiguousarray(np.asarray(flux_poly, dtype=np.float64))
            self._check(self._lib.pdeopt_set_time_terms_poly(self._h, len(th), th.ctypes.data_as(C.c_void_p), len(fl),
                                                             fl.ctypes.data_as(C.c_void_p)))

    def _set_time_thunk(self, fn, const):

        def thunk(t, out, _user):
            a, b, f = fn(t)
            out[0], out[1], out[2] = float(a), float(b), float(f)

        self._time_thunk = L.TIME_FN(thunk)
        self._check(self._lib.pdeopt_set_time_terms(self._h, self._time_thunk, None, const))

    # -- state ------------------------------------------------------------------------------
    def set_state(self, state, env_first: int = 0):
        a = np.ascontiguousarray(np.asarray(state, dtype=self.dtype))
        if a.shape == self.state_shape:
            a = a[None]
        if a.shape[1:] != self.state_shape:
            raise ValueError(f"state shape {a.shape} does not match (batch,)+{self.state_shape}")
        self._check(self._lib.pdeopt_set_state(self._h, int(env_first), a.shape[0], a.ctypes.data_as(C.c_void_p)))

    def get_state(self, env_first: int = 0, env_count: Optional[int] = None, out: Optional[np.ndarray] = None) -> np.ndarray:
        n = self.batch - env_first if env_count is None else env_count
        if out is None:
            out = np.empty((n,) + self.state_shape, dtype=self.dtype)
        elif out.shape != (n,) + self.state_shape or out.dtype != self.dtype or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous array of the state dtype and shape (envs,) + state_shape")
        self._check(self._lib.pdeopt_get_state(self._h, int(env_first), int(n), out.ctypes.data_as(C.c_void_p)))
        return out

    def state_device_ptr(self):
        p, nbytes = C.c_void_p(), C.c_int64()
        self._check(self._lib.pdeopt_state_device_ptr(self._h, C.byref(p), C.byref(nbytes)))
        return p.value, nbytes.value

    # -- compute ----------------------------------------------------------------------------
    def rhs(self, t: float = 0.0, fetch: bool = True) -> Optional[np.ndarray]:
        out = np.empty((self.batch,) + self.state_shape, dtype=self.dtype) if fetch else None
        ptr = out.ctypes.data_as(C.c_void_p) if fetch else None
        self._check(self._lib.pdeopt_rhs(self._h, float(t), ptr))
        return out

    def _upload_time_table(self, integrator: int, dt: float, n: int, t0: float):
        """Fixed-step Euler / RK4 with time-dependent scalar terms (smoothed-boundary theta(t), flux(t)): sample them
        at every stage time of the call up front -- the times formed exactly as the library forms them -- so the
        substep loop runs without a host callback per stage (``pdeopt_set_time_table``)."""
        fn = getattr(self, "_time_terms_fn", None)
        if fn is None or integrator not in (L.INT_EULER, L.INT_RK4) or n <= 0 or n > 100000:
            return
        ts = float(t0) + np.arange(int(n), dtype=np.float64) * float(dt)  # t0 + (double) step * dt
        if integrator == L.INT_RK4:
            ts = np.unique(np.concatenate([ts, ts + float(dt) / 2, ts + float(dt)]))
        terms = np.ascontiguousarray(np.asarray([fn(float(t)) for t in ts], dtype=np.float64).reshape(-1, 3))
        ts = np.ascontiguousarray(ts)
        self._check(self._lib.pdeopt_set_time_table(self._h, len(ts), ts.ctypes.data_as(C.c_void_p), terms.ctypes.data_as(C.c_void_p)))

    def advance(self, integrator: int, dt: float, n_substeps: int, t0: float = 0.0):
        self._upload_time_table(int(integrator), dt, int(n_substeps), t0)
        self._check(self._lib.pdeopt_advance(self._h, int(integrator), float(t0), float(dt), int(n_substeps)))

    def snapshot(self):
        self._check(self._lib.pdeopt_snapshot(self._h))

    def get_interpolated(self, theta: float, env_first: int = 0, env_count: Optional[int] = None):
        n = self.batch - env_first if env_count is None else env_count
        out = np.empty((n,) + self.state_shape, dtype=self.dtype)
        self._check(
            self._lib.pdeopt_get_interpolated(self._h, float(theta), int(env_first), int(n), out.ctypes.data_as(C.c_void_p))
        )
        return out

    def reduce(self, op: int) -> np.ndarray:
        out = np.empty(self.batch, dtype=np.float64)
        self._check(self._lib.pdeopt_reduce(self._h, int(op), out.ctypes.data_as(C.c_void_p)))
        return out

    def probe(self, cells, env_first: int = 0, env_count: Optional[int] = None) -> np.ndarray:
        """State at grid cells ``cells`` ((n, 2) index pairs; (n, 3) for 3-D problems) of every environment of
        the range: ``(envs, n)`` float64, ``(envs, n, 2)`` for the GPE -- a few numbers per environment cross
        PCIe instead of the field."""
        n = self.batch - env_first if env_count is None else env_count
        nd = 3 if (self.problem.nz > 1) else 2
        c = np.ascontiguousarray(np.asarray(cells, dtype=np.int32).reshape(-1, nd))
        comps = 2 if self.problem.equation == L.EQ_GPE else 1
        out = np.empty((n, c.shape[0], comps), dtype=np.float64)
        self._check(self._lib.pdeopt_probe(self._h, c.ctypes.data_as(C.c_void_p), c.shape[0], int(env_first), int(n),
                                           out.ctypes.data_as(C.c_void_p)))
        return out if comps == 2 else out[..., 0]

    def pinned_empty(self, shape, dtype) -> np.ndarray:
        """numpy array over page-locked host memory owned by this engine (``pdeopt_host_alloc``): the target of
        per-step fetches (``observe_u8(out=...)``, ``get_state(out=...)``).  Valid until the engine is closed."""
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        self._check(self._lib.pdeopt_host_alloc(self._h, nbytes, C.byref(p)))
        buf = (C.c_char * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=dt).reshape(shape)

    def observe_u8(self, lo: float, hi: float, env_first: int = 0, env_count: Optional[int] = None,
                   out: Optional[np.ndarray] = None) -> np.ndarray:
        """uint8 frames ``rint(clip((x-lo)/(hi-lo), 0, 1) * 255)`` of shape (envs, nx, ny), quantised on the GPU;
        ``out``: a reusable (ideally ``pinned_empty``) array to write them into"""
        n = self.batch - env_first if env_count is None else env_count
        if out is None:
            out = np.empty((n,) + self.state_shape, dtype=np.uint8)
        elif out.shape != (n,) + self.state_shape or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint8 array of shape (envs, nx, ny)")
        self._check(self._lib.pdeopt_observe_u8(self._h, float(lo), float(hi), int(env_first), int(n),
                                                out.ctypes.data_as(C.c_void_p)))
        return out

    def observe_u8_device(self, lo: float, hi: float, env_first: int = 0, env_count: Optional[int] = None) -> "DeviceArray":
        """The same frames left on the GPU: a ``DeviceArray`` (``__cuda_array_interface__``; ``.torch()`` gives a
        zero-copy ``torch.uint8`` tensor) over a library-owned buffer that the next ``observe_u8*`` call overwrites."""
        n = self.batch - env_first if env_count is None else env_count
        p, nbytes = C.c_void_p(), C.c_int64()
        self._check(self._lib.pdeopt_observe_u8_device(self._h, float(lo), float(hi), int(env_first), int(n),
                                                       C.byref(p), C.byref(nbytes)))
        return DeviceArray(p.value, (n,) + self.state_shape, np.uint8, self.device, owner=self)

    def state_device_array(self) -> "DeviceArray":
        """The state field itself as a ``DeviceArray`` (batch,) + state_shape in the engine's dtype: valid until the
        next ``configure`` with another shape; ``advance`` updates it in place (synchronise with ``sync()``)."""
        ptr, _ = self.state_device_ptr()
        return DeviceArray(ptr, (self.batch,) + self.state_shape, self.dtype, self.device, owner=self)

    def detect_vortices(self, amp_thresh: float = 0.0, tol: float = 0.5, env_first: int = 0,
                        env_count: Optional[int] = None, want_winding: bool = True):
        """Phase-winding census of the resident GPE state (rl_utils.detect_vortices on the GPU).
        Returns ``(counts, winding)``: counts is (envs, 3) int64 = num_vortices, total charge,
        sum |charge|; winding is (envs, nx, ny) int32 or None."""
        n = self.batch - env_first if env_count is None else env_count
        counts = np.zeros((n, 3), dtype=np.int64)
        winding = np.empty((n,) + self.state_shape[:2], dtype=np.int32) if want_winding else None
        self._check(self._lib.pdeopt_detect_vortices(
            self._h, float(amp_thresh), float(tol), int(env_first), int(n),
            winding.ctypes.data_as(C.c_void_p) if want_winding else None, counts.ctypes.data_as(C.c_void_p)))
        return counts, winding

    def tsit5_trial(self, t: float, dt: float, rtol: float, atol: float) -> np.ndarray:
        err = np.empty(self.batch, dtype=np.float64)
        self._check(
            self._lib.pdeopt_tsit5_trial(self._h, float(t), float(dt), float(rtol), float(atol), err.ctypes.data_as(C.c_void_p))
        )
        return err

    def tsit5_trial_env(self, t: float, dts, rtol: float, atol: float):
        """one trial step with a step size per environment (0 leaves an environment alone); returns
        ``(err_norm[batch], dt_ref)`` -- pass ``dt_ref`` to ``tsit5_dense``"""
        d = np.ascontiguousarray(np.asarray(dts, dtype=np.float64))
        if d.shape != (self.batch,):
            raise ValueError(f"one step size per environment: expected shape ({self.batch},), got {d.shape}")
        err = np.empty(self.batch, dtype=np.float64)
        ref = C.c_double()
        self._check(self._lib.pdeopt_tsit5_trial_env(self._h, float(t), d.ctypes.data_as(C.c_void_p), float(rtol),
                                                     float(atol), C.byref(ref), err.ctypes.data_as(C.c_void_p)))
        return err, ref.value

    def tsit5_commit_env(self, accept):
        a = np.ascontiguousarray(np.asarray(accept, dtype=np.uint8))
        if a.shape != (self.batch,):
            raise ValueError(f"one flag per environment: expected shape ({self.batch},), got {a.shape}")
        self._check(self._lib.pdeopt_tsit5_commit_env(self._h, a.ctypes.data_as(C.c_void_p)))

    def tsit5_dense(self, theta: float, dt: float, env_first: int = 0, env_count: Optional[int] = None) -> np.ndarray:
        """4th-order dense output ``y(t + theta dt)`` of the pending Tsit5 trial step (diffrax.Tsit5's interpolant)"""
        n = self.batch - env_first if env_count is None else env_count
        out = np.empty((n,) + self.state_shape, dtype=self.dtype)
        self._check(self._lib.pdeopt_tsit5_dense(self._h, float(theta), float(dt), int(env_first), int(n),
                                                 out.ctypes.data_as(C.c_void_p)))
        return out

    def tsit5_commit(self, accept: bool):
        self._check(self._lib.pdeopt_tsit5_commit(self._h, int(bool(accept))))

    def tsit5_solve_small_supported(self) -> bool:
        """can ``tsit5_solve_small`` take the configured problem? (LDS-resident Cahn-Hilliard / Allen-Cahn FD grids)"""
        return bool(self._lib.pdeopt_tsit5_solve_small_supported(self._h))

    def tsit5_solve_small(self, t0: float, t1: float, dt0: float, controller, max_steps: int, save_ts=()):
        """The whole adaptive Tsit5 solve ``t0 -> t1`` in one launch, one controller per environment
        (``pdeopt_tsit5_solve_small``).  ``controller``: a ``PIDController``; ``save_ts``: ascending times in
        ``(t0, t1]``.  Returns ``(saves, stats)``: saves ``(len(save_ts), batch) + state_shape`` (NaN where an
        environment never got there), stats a list of dicts per environment (t, dt, accepted, rejected, status)."""
        c = controller
        pid = L.Pid(float(c.rtol), float(c.atol), float(c.pcoeff), float(c.icoeff), float(c.dcoeff),
                    -np.inf if c.dtmin is None else float(c.dtmin), np.inf if c.dtmax is None else float(c.dtmax),
                    float(c.factormin), float(c.factormax), float(c.safety))
        ts = np.ascontiguousarray(np.asarray(save_ts, dtype=np.float64))
        saves = np.empty((len(ts), self.batch) + self.state_shape, dtype=self.dtype)
        stats = (L.Tsit5Stats * self.batch)()
        self._check(self._lib.pdeopt_tsit5_solve_small(
            self._h, float(t0), float(t1), float(dt0), C.byref(pid), int(max_steps), len(ts),
            ts.ctypes.data_as(C.c_void_p) if len(ts) else None, saves.ctypes.data_as(C.c_void_p) if len(ts) else None, stats))
        return saves, [dict(t=s.t, dt=s.dt, accepted=s.accepted, rejected=s.rejected, status=s.status, saved=s.saved) for s in stats]

    # -- domain decomposition (padded layout) ---------------------------------------------------
    def set_halo_layout(self, halo: int):
        """0 = periodic field, 4 / 8 = rank-local tile padded by a 4- / 8-cell halo (takes effect at the next
        configure); 8: one halo exchange per RK4 substep instead of two (fused Cahn-Hilliard stage pairs)"""
        self._check(self._lib.pdeopt_set_option(self._h, L.OPT_HALO_LAYOUT, int(halo)))
        self._aux_keys.clear()  # a layout change re-allocates on the next configure

    def halo_strip_elems(self) -> int:
        v = C.c_int64()
        self._check(self._lib.pdeopt_halo_strip_elems(self._h, C.byref(v)))
        return v.value

    def halo_pack(self, field: int, dev_send: Optional[int] = None):
        self._check(self._lib.pdeopt_halo_pack(self._h, int(field), C.c_void_p(dev_send) if dev_send else None))

    def halo_unpack(self, field: int, dev_recv: Optional[int], neighbours: Sequence[int]):
        nb = (C.c_int * 8)(*[int(v) for v in neighbours])
        self._check(self._lib.pdeopt_halo_unpack(self._h, int(field), C.c_void_p(dev_recv) if dev_recv else None, nb))

    def rk4_phase_plan(self):
        f = (C.c_int * 4)()
        n = C.c_int()
        self._check(self._lib.pdeopt_rk4_phase_plan(self._h, f, C.byref(n)))
        return [f[i] for i in range(n.value)]

    def rk4_phase(self, phase: int, dt: float, part: int = 0):
        """one phase of an RK4 substep on a padded tile; ``part``: 0 all tiles, 1 interior tiles (no halo reads),
        2 edge tiles"""
        self._check(self._lib.pdeopt_rk4_phase_part(self._h, int(phase), float(dt), int(part)))

    def rk4_loopback_advance(self, dt: float, n_substeps: int):
        """n substeps of a single-rank padded tile, loop-back exchange included, in one library call"""
        self._check(self._lib.pdeopt_rk4_loopback_advance(self._h, float(dt), int(n_substeps)))

    # -- RCCL communicator owned by the library (decomposed driver without a host round trip per substep) ----
    def comm_unique_id(self) -> bytes:
        """rank 0: the 128-byte id every rank passes to ``comm_init`` (carry it with any transport)"""
        buf = C.create_string_buffer(128)
        rc = self._lib.pdeopt_comm_unique_id(buf)
        if rc != L.OK:
            raise L.PdeoptError(rc, self._lib.pdeopt_last_error(None).decode())
        return buf.raw

    def comm_init(self, world: int, rank: int, unique_id: bytes):
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        self._check(self._lib.pdeopt_comm_init(self._h, int(world), int(rank), unique_id))

    def comm_init_local(self, group: "LocalGroup", rank: int):
        """join an in-process group of ranks (several engines of this process, on one GPU or one per GPU) instead of
        an RCCL communicator: ``rk4_decomposed_advance`` then exchanges by device-side copies; every rank's thread
        must call it concurrently"""
        self._check(self._lib.pdeopt_comm_init_local(self._h, group._h, int(rank)))
        self._group = group  # the group outlives its members

    def comm_ipc_export(self, world: int, rank: int) -> bytes:
        """peer-mapped exchange (no collective in the substep): allocate this rank's strip buffers + counters and return
        the 64-byte hipIpc handle the other processes map (``comm_ipc_attach``); call after ``configure`` in the
        halo-8 layout"""
        buf = C.create_string_buffer(64)
        self._check(self._lib.pdeopt_comm_ipc_export(self._h, int(world), int(rank), buf))
        return buf.raw

    def comm_ipc_attach(self, handles: Sequence[bytes]):
        """map every rank's block: ``handles[r]`` = what rank r's ``comm_ipc_export`` returned (own entry ignored)"""
        blob = b"".join(bytes(h) for h in handles)
        if len(blob) % 64:
            raise ValueError("hipIpc handles are 64 bytes each")
        self._check(self._lib.pdeopt_comm_ipc_attach(self._h, blob))

    def comm_destroy(self):
        self._check(self._lib.pdeopt_comm_destroy(self._h))

    def rk4_decomposed_advance(self, dt: float, n_substeps: int, neighbours: Sequence[int], overlap: bool = True):
        nb = (C.c_int * 8)(*[int(v) for v in neighbours])
        self._check(self._lib.pdeopt_rk4_decomposed_advance(self._h, float(dt), int(n_substeps), nb, int(bool(overlap))))

    def buffer_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self._lib.pdeopt_buffer_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def buffer_free(self, ptr: int):
        self._check(self._lib.pdeopt_buffer_free(self._h, C.c_void_p(ptr)))

    def buffer_copy(self, dst, src, nbytes: int, kind: int):
        as_ptr = lambda v: v.ctypes.data_as(C.c_void_p) if isinstance(v, np.ndarray) else C.c_void_p(int(v))
        self._check(self._lib.pdeopt_buffer_copy(self._h, as_ptr(dst), as_ptr(src), int(nbytes), int(kind)))

    def stage_launches(self) -> int:
        v = C.c_int64()
        self._check(self._lib.pdeopt_get_counter(self._h, L.CNT_STAGE_LAUNCHES, C.byref(v)))
        return v.value

    def last_groups(self) -> int:
        """environment groups the last ``advance`` ran the batch in (1 = the whole batch per sweep)"""
        v = C.c_int64()
        self._check(self._lib.pdeopt_get_counter(self._h, L.CNT_LAST_GROUPS, C.byref(v)))
        return v.value

    def last_group_streams(self) -> int:
        """groups the last ``advance`` kept in flight side by side on separate HIP streams (1 or 2)"""
        v = C.c_int64()
        self._check(self._lib.pdeopt_get_counter(self._h, L.CNT_GROUP_STREAMS, C.byref(v)))
        return v.value

    def sync(self):
        self._check(self._lib.pdeopt_sync(self._h))

    def timer_start(self):
        self._check(self._lib.pdeopt_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_double()
        self._check(self._lib.pdeopt_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def timer_clock_hz(self) -> float:
        """shader clock held between the last ``timer_start`` / ``timer_stop`` (s_memtime over s_memrealtime); 0.0 if unmeasured"""
        hz = C.c_double()
        self._check(self._lib.pdeopt_timer_clock(self._h, C.byref(hz)))
        return hz.value


class LocalGroup:
    """``pdeopt_local_group``: an in-process group of ``world`` ranks for the decomposed driver (virtual ranks on one
    GPU, or one engine per GPU with a host thread each).  Keep it alive as long as any member engine."""

    def __init__(self, world: int):
        self._lib = L.load_library()
        h = C.c_void_p()
        rc = self._lib.pdeopt_local_group_create(int(world), C.byref(h))
        if rc != L.OK:
            raise L.PdeoptError(rc, self._lib.pdeopt_last_error(None).decode())
        self._h, self.world = h, int(world)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pdeopt_local_group_destroy(self._h)
            self._h = None


_default_engines: dict = {}


class DeviceArray:
    """A typed view of library-owned device memory, exported through ``__cuda_array_interface__`` (version 2) --
    what PyTorch-ROCm, CuPy and Numba read: the zero-copy hand-over to a consumer on the same GPU."""

    def __init__(self, ptr: int, shape, dtype, device: int, owner=None):
        self.ptr, self.shape, self.dtype, self.device, self._owner = int(ptr), tuple(int(v) for v in shape), np.dtype(dtype), int(device), owner

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2, "strides": None}

    def torch(self):
        import torch

        return torch.as_tensor(self, device=torch.device("cuda", self.device))


def default_engine(device: int = 0) -> HipEngine:
    """Process-wide engine per device, shared by equation objects that only need ``rhs``."""
    eng = _default_engines.get(device)
    if eng is None:
        eng = _default_engines[device] = HipEngine(device)
    return eng
