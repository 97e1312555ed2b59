"""Allen-Cahn and Cahn-Hilliard with the smoothed-boundary method (arbitrary geometry through a
level-set field psi), 2-D finite differences, on the HIP engine.

Same dataclass surface as the reference (pde_opt/numerics/equations/allen_cahn.py:88-159,
cahn_hilliard.py:204-289): fields ``domain, kappa, f, mu, R|D, theta[, flux], derivs``; published
attributes ``psi, sqrt_kappa, hx, hy, norm_grad_psi, left_half``; ``rhs(state, t)``.
``psi = domain.geometry.smooth`` -- any object with a ``smooth`` array works as ``geometry`` (the
reference's ``Shape`` builds it by relaxing a binary mask; that preprocessing is outside the hot
path, SURVEY section 8 out-of-scope list).

The stencil kernel (csrc/stencil_generic.hpp, SBM cases) evaluates

    inner = mu_h(u) - kappa/psi div(psi grad u) - sqrt(kappa) |grad psi|/psi sqrt(2 f(u)) w(t)
    AC:  du/dt = -R(u) inner                       w = cos(theta) left_half
    CH:  du/dt = div(psi D(u) grad inner)/psi + |grad psi|/psi flux(t)
                                                   w = cos(theta) left_half + cos(pi - theta)(1 - left_half)

``theta(t)`` and ``flux(t)`` are ordinary Python callables of time; the library calls back for
their values at every Runge-Kutta stage time (``pdeopt_set_time_terms``), so time-dependent contact
angles (notebooks/smooth_boundary.ipynb) integrate exactly as upstream.
"""

from __future__ import annotations

import dataclasses
from typing import Any

import numpy as np

from ... import _lib as L
from ..closures import as_closure
from ..domains import Domain
from .base_eq import BaseEquation


def _norm_grad_over_psi(psi, hx, hy):
    """sqrt(gradx_c(psi)^2 + grady_c(psi)^2) / psi with periodic centred differences
    (allen_cahn.py:128-133; derivatives.py:69-76)"""
    gx = 0.5 * (np.roll(psi, -1, 0) - np.roll(psi, 1, 0)) / hx
    gy = 0.5 * (np.roll(psi, -1, 1) - np.roll(psi, 1, 1)) / hy
    return np.sqrt(gx**2 + gy**2) / psi


class _SmoothedBoundary(BaseEquation):
    _equation_code = -1

    def rhs(self, state, t):  # replaced in __post_init__, as upstream
        raise NotImplementedError("rhs method not implemented")

    def _init_geometry(self, name):
        if len(self.domain.points) != 2:
            raise ValueError(f"{name} needs a 2-D domain")
        if self.domain.geometry is None or not hasattr(self.domain.geometry, "smooth"):
            raise ValueError(f"{name} needs domain.geometry.smooth (the level-set field psi)")
        self.psi = np.asarray(self.domain.geometry.smooth, dtype=np.float64)
        if self.psi.shape != tuple(self.domain.points):
            raise ValueError(f"psi shape {self.psi.shape} does not match domain points {self.domain.points}")
        self.sqrt_kappa = np.sqrt(self.kappa)
        self.hx, self.hy = self.domain.dx
        self.norm_grad_psi = _norm_grad_over_psi(self.psi, self.hx, self.hy)
        self.left_half = np.zeros_like(self.psi)
        self._f_desc = as_closure(self.f)
        self._mu_desc = as_closure(self.mu)
        if self.derivs == "fd":
            self.rhs = self.rhs_fd
        else:
            raise ValueError(f"Invalid derivative type: {self.derivs}")

    def _engine_problem(self):
        nx, ny = self.domain.points
        return dict(equation=self._equation_code, nx=nx, ny=ny, hx=self.hx, hy=self.hy,
                    kappa=float(self.kappa), mu=self._mu_desc, mob=self._mob_desc, fe=self._f_desc,
                    derivs=L.DERIVS_FD)

    def _engine_upload(self, engine, t: float = 0.0, t_end=None):
        engine.set_aux(L.AUX_SBM_PSI, self.psi)
        engine.set_aux(L.AUX_SBM_NORM_GRAD, self.norm_grad_psi)
        engine.set_aux(L.AUX_SBM_MASK, self.left_half)
        # theta(t), flux(t) as polynomials where they are ones (constants included): the in-kernel adaptive solve
        # chooses its stage times on the device and evaluates them there
        from ..closures import poly_in_t

        th, fl = poly_in_t(self.theta), poly_in_t(getattr(self, "flux", 0.0))  # (Allen-Cahn has no boundary flux)
        if th is not None and fl is not None:
            engine.set_time_terms(self._time_terms, theta_poly=th, flux_poly=fl)
        else:
            engine.set_time_terms(self._time_terms)

    def _time_dependent_rhs(self, t0: float = 0.0, t1=None) -> bool:
        return True  # theta(t) / flux(t) are evaluated at every stage time

    def rhs_fd(self, state, t):
        return self._run_rhs(state, t)


@dataclasses.dataclass
class AllenCahn2DSmoothedBoundary(_SmoothedBoundary):
    """du/dt = -R(u) (mu_h(u) - kappa/psi div(psi grad u) - sqrt(kappa)|grad psi|/psi sqrt(2f) cos(theta))."""

    domain: Domain
    kappa: float
    f: Any
    mu: Any
    R: Any
    theta: Any
    derivs: str = "fd"

    _equation_code = L.EQ_ALLEN_CAHN_SBM
    # the kernels read kappa and the closure coefficients from the per-environment table (theta is shared)
    _per_env_controls = frozenset({"kappa", "mu", "R"})

    def __post_init__(self):
        self._init_geometry("AllenCahn2DSmoothedBoundary")
        self.left_half[:, :100] = 1.0  # allen_cahn.py:135 (hard-coded upstream)
        self._mob_desc = as_closure(self.R)

    def _time_terms(self, t):
        return np.cos(self.theta(t)), 0.0, 0.0


@dataclasses.dataclass
class CahnHilliard2DSmoothedBoundary(_SmoothedBoundary):
    """du/dt = div(psi D(u) grad(inner))/psi + |grad psi|/psi flux(t)."""

    domain: Domain
    kappa: float
    f: Any
    mu: Any
    D: Any
    theta: Any
    flux: Any
    derivs: str = "fd"

    _equation_code = L.EQ_CAHN_HILLIARD_SBM
    _per_env_controls = frozenset({"kappa", "mu", "D"})

    def __post_init__(self):
        self._init_geometry("CahnHilliard2DSmoothedBoundary")
        self.left_half[:50, :] = 1.0  # cahn_hilliard.py:254 (hard-coded upstream)
        self._mob_desc = as_closure(self.D)

    def _time_terms(self, t):
        th = self.theta(t)
        return np.cos(th), np.cos(np.pi - th), self.flux(t)
