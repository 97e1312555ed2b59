"""Test-only stand-in for HipEngine, backed by the numpy oracle, so the HOST logic of the product
(pde_opt_amd/integrate.py: step plans, SaveAt interpolation, the PID controller; pde_env.py protocol)
runs in the CPU suite.  It implements exactly the engine methods those modules call.  It lives under
tests/ and is injected through the public ``engine=`` arguments; the product never imports it."""
import numpy as np

from oracle import np_oracle as O
from pde_opt_amd import _lib as L


class OracleEngine:
    last_kernel = "oracle(test double)"

    def __init__(self):
        self.problem = None
        self.calls = []

    # -- configuration ------------------------------------------------------------------------
    def configure(self, equation, dtype, nx, ny, batch, hx, hy, kappa=0.0, mu=None, mob=None, gpe_k=0.0, derivs=0, fe=None,
                  nz=0, hz=0.0):
        same = self.problem is not None and (self.eq, self.dtype, self.batch, self.state_shape[:2]) == (
            equation, np.dtype(dtype), batch, (nx, ny))
        self.eq, self.dtype, self.batch = equation, np.dtype(dtype), batch
        self.hx, self.hy, self.kappa, self.mu, self.mob = hx, hy, kappa, mu, mob
        self.problem = object()
        self.state_shape = (nx, ny) + ((2,) if equation == L.EQ_GPE else ())
        if not same:  # a same-shape configure keeps the resident state and the aux fields, like the library
            self.y = np.zeros((batch,) + self.state_shape, self.dtype)
            self.aux, self.aux_fn, self.spots = {}, {}, None
            self.imex_A, self.symbol = 0.5, None
            self.sbm, self.time_fn = {}, None
        self.fe = fe
        self.kappa_env = np.full(batch, float(kappa))
        self.gpe_k_env = np.full(batch, float(gpe_k))
        self.imex_sigma = np.ones(batch)
        self.time_scale, self.strang_dx = 1.0, 1.0

    def set_aux(self, which, field, per_env=False, key=None):
        self.aux[which] = (np.asarray(field), bool(per_env))
        self.aux_fn.pop(which, None)  # a static upload replaces a time-dependent source
        if which == L.AUX_IMEX_SYMBOL:
            self.symbol = np.asarray(field)
        if which in (L.AUX_SBM_PSI, L.AUX_SBM_NORM_GRAD, L.AUX_SBM_MASK):
            self.sbm[which] = np.asarray(field)

    def set_aux_time_fn(self, which, fn, per_env=False):
        if fn is None:
            self.aux_fn.pop(which, None)
        else:
            self.aux_fn[which] = (fn, bool(per_env))
            self.calls.append(("aux_time_fn", which, bool(per_env)))

    def _aux_at(self, which, t, b):
        """aux field `which` of environment b at local time t (the library refreshes per substep / stage)"""
        if which in self.aux_fn:
            fn, per_env = self.aux_fn[which]
            self.calls.append(("aux_eval", which, t))
            a = np.asarray(fn(t))
        else:
            a, per_env = self.aux[which]
        return a[b] if per_env else a

    def set_gpe_spots(self, tables, x_first=0.0, y_first=0.0, env_first=0):
        self.calls.append(("gpe_spots", None if tables is None else np.asarray(tables).shape))
        if tables is None:
            self.spots = None
            return
        t = np.asarray(tables, dtype=float)
        if getattr(self, "spots", None) is None or self.spots.shape[1:] != t.shape[1:]:
            self.spots = np.zeros((self.batch,) + t.shape[1:])
        self.spots[env_first:env_first + t.shape[0]] = t
        nx, ny = self.state_shape[:2]
        self.spot_mesh = np.meshgrid(x_first + self.hx * np.arange(nx), y_first + self.hy * np.arange(ny), indexing="ij")

    def _spots_at(self, t, b):
        """the ABI's expression (include/pdeopt_hip.h, pdeopt_set_gpe_spots)"""
        if getattr(self, "spots", None) is None:
            return 0.0
        X, Y = self.spot_mesh
        w = 0.0
        for a0, a1, x0, x1, y0, y1, c in self.spots[b]:
            w = w + (a0 + a1 * t) * np.exp(-((X - x0 - x1 * t) ** 2 + (Y - y0 - y1 * t) ** 2) * c)
        return w

    def set_env_imex_scale(self, env_first, sigma):
        sigma = np.atleast_1d(np.asarray(sigma, dtype=float))
        self.imex_sigma[env_first:env_first + len(sigma)] = sigma

    def set_env_gpe_k(self, env_first, k):
        k = np.atleast_1d(np.asarray(k, dtype=float))
        self.gpe_k_env[env_first:env_first + len(k)] = k

    def set_env_params(self, env_first, kappa=None, mu_coef=None, mob_coef=None):
        if kappa is not None:
            kappa = np.atleast_1d(np.asarray(kappa, dtype=float))
            self.kappa_env[env_first:env_first + len(kappa)] = kappa

    def set_time_terms(self, fn=None, constant=(0.0, 0.0, 0.0), theta_poly=None, flux_poly=None):
        self.time_fn = fn if fn is not None else (lambda t: constant)

    def set_integrator_params(self, imex_A=0.5, time_scale=1.0, strang_dx=1.0):
        self.imex_A, self.time_scale, self.strang_dx = imex_A, time_scale, strang_dx

    def set_state(self, state, env_first=0):
        a = np.asarray(state, dtype=self.dtype)
        a = a[None] if a.ndim == len(self.state_shape) else a
        self.y[env_first:env_first + a.shape[0]] = a

    def get_state(self, env_first=0, env_count=None):
        n = self.batch - env_first if env_count is None else env_count
        return self.y[env_first:env_first + n].copy()

    # -- compute --------------------------------------------------------------------------------
    def _f(self, t, u, b=0):
        if self.eq == L.EQ_ADVECTION_DIFFUSION:
            return O.ad_rhs_fd(u, self.hx, self.hy, self._aux_at(L.AUX_VX_FACE, t, b), self._aux_at(L.AUX_VY_FACE, t, b),
                               self.kappa_env[b])
        if self.eq in (L.EQ_ALLEN_CAHN_SBM, L.EQ_CAHN_HILLIARD_SBM):
            # the ABI's decomposition (include/pdeopt_hip.h, pdeopt_time_fn): scalars from the
            # callback, spatial fields from the aux uploads
            a, b, fl = self.time_fn(t)
            psi, ngp, m = (self.sbm[k] for k in (L.AUX_SBM_PSI, L.AUX_SBM_NORM_GRAD, L.AUX_SBM_MASK))
            inner = O.sbm_inner(u, psi, self.hx, self.hy, self.kappa, self.fe, self.mu, a * m + b * (1 - m))
            if self.eq == L.EQ_ALLEN_CAHN_SBM:
                return -self.mob(u) * inner
            Du = self.mob(u)
            Fx = O.avg_face(psi, 0) * O.avg_face(Du, 0) * O.grad_face(inner, self.hx, 0)
            Fy = O.avg_face(psi, 1) * O.avg_face(Du, 1) * O.grad_face(inner, self.hy, 1)
            return (O.div_face(Fx, self.hx, 0) + O.div_face(Fy, self.hy, 1)) / psi + ngp * fl
        if self.eq == L.EQ_SHAPE_SMOOTH:
            return O.shape_smooth_rhs(u, self.hx, self.hy, self.gpe_k_env[b], self.kappa_env[b])
        fn = O.ch_rhs_fd if self.eq == L.EQ_CAHN_HILLIARD else O.ac_rhs_fd
        return fn(u, self.hx, self.hy, self.kappa_env[b], self.mu, self.mob)

    def rhs(self, t=0.0, fetch=True):
        out = np.stack([self._f(t, self.y[b].astype(np.float64), b) for b in range(self.batch)]).astype(self.dtype)
        return out if fetch else None

    def advance(self, integrator, dt, n, t0=0.0):
        self.calls.append(("advance", integrator, dt, n, t0))
        for b in range(self.batch):
            u = self.y[b]
            f = lambda t, v, b=b: self._f(t, v, b)
            for i in range(int(n)):
                t = t0 + i * dt
                if integrator == L.INT_EULER:
                    u = O.euler_step(f, t, u, dt)
                elif integrator == L.INT_RK4:
                    u = O.rk4_step(f, t, u, dt)
                elif integrator == L.INT_TSIT5:
                    u = O.tsit5_step(f, t, u, dt)[0]
                elif integrator == L.INT_IMEX:
                    u = O.imex_step(f, t, u, dt, self.imex_A, self.imex_sigma[b] * self.symbol)
                elif integrator == L.INT_STRANG:
                    # b = -i (V(t0) + k |psi0|^2): the ABI's decomposition of gross_pitaevskii.py:67-75
                    def bterm(tt, yy, b=b):
                        w = (self._aux_at(L.AUX_GPE_POTENTIAL, tt, b) + self._spots_at(tt, b)
                             + self.gpe_k_env[b] * (yy[..., 0] ** 2 + yy[..., 1] ** 2))
                        return np.stack([np.zeros_like(w), -w], axis=-1)

                    u = O.strang_step(bterm, t, u, dt, self.aux[L.AUX_GPE_A_TERM][0], self.strang_dx, self.time_scale)
                else:
                    raise ValueError(integrator)
            self.y[b] = u

    def close(self):
        self.closed = True

    # -- on-device rewards / observations, by the ABI's definitions (include/pdeopt_hip.h) --------------
    def reduce(self, op):
        y = self.y.reshape(self.batch, -1).astype(np.float64)
        return {L.RED_MEAN: y.mean(1), L.RED_VAR: y.var(1), L.RED_MIN: y.min(1), L.RED_MAX: y.max(1),
                L.RED_SUMSQ: (y**2).sum(1), L.RED_NONFINITE: (~np.isfinite(y)).sum(1).astype(np.float64)}[op]

    def observe_u8(self, lo, hi, env_first=0, env_count=None, out=None):
        n = self.batch - env_first if env_count is None else env_count
        q = np.rint(np.clip((self.y[env_first:env_first + n].astype(np.float64) - lo) / (hi - lo), 0.0, 1.0) * 255.0).astype(np.uint8)
        if out is not None:
            out[...] = q
            return out
        return q

    def probe(self, cells, env_first=0, env_count=None):
        n = self.batch - env_first if env_count is None else env_count
        c = np.asarray(cells, dtype=int).reshape(-1, 2)
        return np.stack([self.y[b][c[:, 0], c[:, 1]].astype(np.float64) for b in range(env_first, env_first + n)])

    def snapshot(self):
        self.snap = self.y.copy()

    def get_interpolated(self, theta, env_first=0, env_count=None):
        n = self.batch - env_first if env_count is None else env_count
        s = slice(env_first, env_first + n)
        return self.snap[s] + theta * (self.y[s] - self.snap[s])

    def tsit5_trial(self, t, dt, rtol, atol):
        self.pending, self.pending_ks, errs = [], [], []
        for b in range(self.batch):
            y1, err, _, ks = O.tsit5_step(lambda tt, v, b=b: self._f(tt, v, b), t, self.y[b], dt, return_slopes=True)
            self.pending_ks.append(ks)
            sc = atol + rtol * np.maximum(np.abs(self.y[b]), np.abs(y1))
            errs.append(np.sqrt(np.mean((err / sc) ** 2)))
            self.pending.append(y1)
        return np.asarray(errs)

    def tsit5_trial_env(self, t, dts, rtol, atol):
        """the ABI's scheme: slopes scaled by dt_b / dt_ref, shared coefficients dt_ref a_ij"""
        dts = np.asarray(dts, dtype=float)
        ref = float(dts.max())
        self.pending, self.pending_ks, errs = [], [], []
        for b in range(self.batch):
            sc = dts[b] / ref
            y1, err, _, ks = O.tsit5_step(lambda tt, v, b=b, sc=sc: sc * self._f(tt, v, b), t, self.y[b], ref, return_slopes=True)
            self.pending_ks.append(ks)
            scale = atol + rtol * np.maximum(np.abs(self.y[b]), np.abs(y1))
            errs.append(np.sqrt(np.mean((err / scale) ** 2)))
            self.pending.append(y1)
        return np.asarray(errs), ref

    def tsit5_commit_env(self, accept):
        for b, a in enumerate(accept):
            if a:
                self.y[b] = self.pending[b]

    def tsit5_dense(self, theta, dt, env_first=0, env_count=None):
        n = self.batch - env_first if env_count is None else env_count
        return np.stack([O.tsit5_dense(self.y[b], dt, self.pending_ks[b], theta) for b in range(env_first, env_first + n)])

    def tsit5_commit(self, accept):
        if accept:
            self.y = np.stack(self.pending)

    # -- the in-kernel adaptive solve (pdeopt_tsit5_solve_small), emulated: one controller per environment ------------
    small_adaptive = False  # set True to route diffeqsolve through integrate._solve_adaptive_in_kernel on the CPU

    def tsit5_solve_small_supported(self):
        return bool(self.small_adaptive)

    def tsit5_solve_small(self, t0, t1, dt0, controller, max_steps, save_ts=()):
        """what csrc/stencil_small_adaptive.hpp does per workgroup, in numpy: the statements of the kernel's loop in
        the kernel's order (FSAL slope, min(dt, t1 - t), stall / budget exits, PID factor, dense output, clipping)"""
        from pde_opt_amd.integrate import _clip_dt, _pid_update

        c = controller
        ts = [float(v) for v in save_ts]
        saves = np.full((len(ts), self.batch) + self.state_shape, np.nan, dtype=self.dtype)
        stats = []
        for b in range(self.batch):
            f = lambda tt, v, b=b: self._f(tt, v, b)
            y, t, dt = self.y[b].astype(np.float64), float(t0), float(dt0)
            k1 = f(t, y)
            prev = pprev = 1.0
            acc = rej = qi = status = 0
            while t < t1:
                if acc + rej >= max_steps:
                    status = L.TSIT5_MAX_STEPS
                    break
                h = min(dt, t1 - t)
                if not (h > 0.0) or t + h == t:
                    status = L.TSIT5_STALLED
                    break
                y1, err, k7, ks = O.tsit5_step(f, t, y, h, k1=k1, return_slopes=True)
                sc = c.atol + c.rtol * np.maximum(np.abs(y), np.abs(y1))
                en = float(np.sqrt(np.mean((err / sc) ** 2)))
                keep, fac, inv = _pid_update(c, en, prev, pprev)
                if keep:
                    acc += 1
                    t_new = t + h
                    while qi < len(ts) and ts[qi] <= t_new + 1e-14 * max(1.0, abs(t_new)):
                        saves[qi, b] = O.tsit5_dense(y, h, ks, min(1.0, max(0.0, (ts[qi] - t) / h)))
                        qi += 1
                    y, k1 = y1, k7
                    t = t_new if t_new < t1 - 1e-14 * max(1.0, abs(t1)) else t1
                    pprev, prev = prev, inv
                else:
                    rej += 1
                dt = _clip_dt(c, h * fac)
            self.y[b] = y
            stats.append(dict(t=t, dt=dt, accepted=acc, rejected=rej, status=status, saved=qi))
        return saves, stats
