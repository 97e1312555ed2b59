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
    def configure(self, equation, dtype, nx, ny, batch, hx, hy, kappa=0.0, mu=None, mob=None, gpe_k=0.0, derivs=0, fe=None):
        self.eq, self.dtype, self.batch = equation, np.dtype(dtype), batch
        self.hx, self.hy, self.kappa, self.mu, self.mob = hx, hy, kappa, mu, mob
        self.problem = object()
        self.state_shape = (nx, ny)
        self.y = np.zeros((batch, nx, ny), self.dtype)
        self.imex_A, self.symbol = 0.5, None
        self.fe, self.sbm, self.time_fn = fe, {}, None

    def set_aux(self, which, field, per_env=False):
        if which == L.AUX_IMEX_SYMBOL:
            self.symbol = np.asarray(field)
        if which in (L.AUX_SBM_PSI, L.AUX_SBM_NORM_GRAD, L.AUX_SBM_MASK):
            self.sbm[which] = np.asarray(field)

    def set_time_terms(self, fn=None, constant=(0.0, 0.0, 0.0)):
        self.time_fn = fn if fn is not None else (lambda t: constant)

    def set_integrator_params(self, imex_A=0.5, time_scale=1.0, strang_dx=1.0):
        self.imex_A = imex_A

    def set_state(self, state, env_first=0):
        a = np.asarray(state, dtype=self.dtype)
        a = a[None] if a.ndim == 2 else a
        self.y[env_first:env_first + a.shape[0]] = a

    def get_state(self, env_first=0, env_count=None):
        n = self.batch - env_first if env_count is None else env_count
        return self.y[env_first:env_first + n].copy()

    # -- compute --------------------------------------------------------------------------------
    def _f(self, t, u):
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
        fn = O.ch_rhs_fd if self.eq == L.EQ_CAHN_HILLIARD else O.ac_rhs_fd
        return fn(u, self.hx, self.hy, self.kappa, self.mu, self.mob)

    def advance(self, integrator, dt, n, t0=0.0):
        self.calls.append(("advance", integrator, dt, n, t0))
        for b in range(self.batch):
            u = self.y[b]
            for i in range(int(n)):
                t = t0 + i * dt
                if integrator == L.INT_EULER:
                    u = O.euler_step(self._f, t, u, dt)
                elif integrator == L.INT_RK4:
                    u = O.rk4_step(self._f, t, u, dt)
                elif integrator == L.INT_IMEX:
                    u = O.imex_step(self._f, t, u, dt, self.imex_A, self.symbol)
                else:
                    raise ValueError(integrator)
            self.y[b] = u

    def snapshot(self):
        self.snap = self.y.copy()

    def get_interpolated(self, theta, env_first=0, env_count=None):
        n = self.batch - env_first if env_count is None else env_count
        s = slice(env_first, env_first + n)
        return self.snap[s] + theta * (self.y[s] - self.snap[s])

    def tsit5_trial(self, t, dt, rtol, atol):
        self.pending, errs = [], []
        for b in range(self.batch):
            y1, err, _ = O.tsit5_step(self._f, t, self.y[b], dt)
            sc = atol + rtol * np.maximum(np.abs(self.y[b]), np.abs(y1))
            errs.append(np.sqrt(np.mean((err / sc) ** 2)))
            self.pending.append(y1)
        return np.asarray(errs)

    def tsit5_commit(self, accept):
        if accept:
            self.y = np.stack(self.pending)
